set -e
ROOT=$(pwd); export TMPDIR=/tmp
python -m pytest tests/test_gpu_cfg5.py tests/test_gpu_ntt.py -x -q -k "cfg5 or wide or keccak_air or configs4 or test_" 2>&1 | tail -2
python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5', round(d['value'],1), 'LDE', round(d['roofline']['avg_us']))"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/r03_p_fetch -- python3 $ROOT/tools/pmc_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/gpurun_out/r03_p_write -- python3 $ROOT/tools/pmc_probe.py > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_summarize.py gpurun_out/r03_p_fetch gpurun_out/r03_p_write gpurun_out/r03_p_pmc_lde.json
python3 -c "
import json; d=json.load(open('gpurun_out/r03_p_pmc_lde.json'))
for l in d['cfg5_lde_2^16x2633_blowup2']['launches']: print(l['kernel'][:50], round(l['fetch_bytes']/1e6), round(l['write_bytes']/1e6))"
