#!/bin/bash
# Counter passes behind bench.py's valu_roofline; run on the GPU box from the repo root:
#   bash tools/run_pmc_poseidon2.sh   ->  gpurun_out/pmc_p2*, profiles/r02_pmc_poseidon2.json
set -e
ROOT=$(pwd)
export TMPDIR=/tmp
CTRS="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
rm -rf gpurun_out/pmc_p2 gpurun_out/pmc_p2_cal
(cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_p2_cal -- $ROOT/tools/_bin/clock_probe > $ROOT/gpurun_out/pmc_p2_cal.log 2>&1)
(cd /tmp && PYTHONPATH=$ROOT rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $ROOT/gpurun_out/pmc_p2 -- python3 $ROOT/tools/pmc_poseidon2_probe.py > $ROOT/gpurun_out/pmc_p2.log 2>&1)
python3 tools/pmc_poseidon2_summarize.py gpurun_out/pmc_p2 gpurun_out/pmc_p2_cal profiles/r02_pmc_poseidon2.json | tee gpurun_out/pmc_p2_summary.txt
cp profiles/r02_pmc_poseidon2.json gpurun_out/r02_pmc_poseidon2.json
