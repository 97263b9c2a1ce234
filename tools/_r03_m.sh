set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/r03_pmc_fetch -- python3 $ROOT/tools/pmc_probe.py > $ROOT/gpurun_out/r03_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/r03_pmc_write -- python3 $ROOT/tools/pmc_probe.py > $ROOT/gpurun_out/r03_pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_summarize.py gpurun_out/r03_pmc_fetch gpurun_out/r03_pmc_write gpurun_out/r03_pmc_lde.json
python3 -c "
import json; d=json.load(open('gpurun_out/r03_pmc_lde.json'))
for k in ('cfg3_lde_2^24x2_blowup4','cfg2_lde_2^20x2_blowup2'):
    for l in d[k]['launches']: print(k, l['kernel'][:40], round(l['fetch_bytes']/1e6,1), round(l['write_bytes']/1e6,1))
"
