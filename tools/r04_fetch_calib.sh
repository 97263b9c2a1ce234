# FETCH_SIZE calibration for the wide LDE's K3 access shape: tools/fetch_calib under rocprofv3 --pmc FETCH_SIZE
set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/r04_fetch_calib -- $ROOT/tools/_bin/fetch_calib > $ROOT/gpurun_out/r04_fetch_calib.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json, re
log = open("gpurun_out/r04_fetch_calib.log").read()
m = re.search(r"bytes: stream16 (\d+)\s+tiles4_w2633 (\d+)\s+tiles4_w2688 (\d+)\s+tiles16_w2688 (\d+)", log)
known = [int(v) for v in m.groups()]
path = glob.glob("gpurun_out/r04_fetch_calib/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
names = ["stream16", "tiles4_w2633", "tiles4_w2688", "tiles16_w2688"]
out = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/_bin/fetch_calib (tools/fetch_calib.hip); second repetition; FETCH_SIZE in the counter's unit (KiB -> bytes x 1024 when the value is small)", "kernels": {}}
k = [r for r in rows if "stream16" in r["Kernel_Name"] or "tiles" in r["Kernel_Name"]][-4:]
for name, r, b in zip(names, k, known):
    v = float(r["Counter_Value"])
    fetched = v * 1024 if v * 1024 < 8 * b else v  # rocprofv3 reports FETCH_SIZE in KiB
    out["kernels"][name] = {"kernel": r["Kernel_Name"][:40], "known_bytes": b, "FETCH_SIZE_bytes": fetched, "factor_to_apply": b / fetched}
    print("%-14s known %.3f GB  FETCH_SIZE %.3f GB  -> multiply FETCH_SIZE by %.3f" % (name, b / 1e9, fetched / 1e9, b / fetched))
json.dump(out, open("gpurun_out/r04_fetch_calib.json", "w"), indent=1)
PY
