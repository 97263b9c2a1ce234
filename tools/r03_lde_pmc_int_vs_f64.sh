set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
for m in 0 7; do
  P3HIP_NTT_NARROW_F64=$m rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_c_trace_$m -o t -- python3 $ROOT/tools/lde_probe.py 20:2:1 22:2:1 22:4:2 10 > $ROOT/gpurun_out/r03_c_trace_$m.log 2>&1
  P3HIP_NTT_NARROW_F64=$m rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/r03_c_pmc_$m -- python3 $ROOT/tools/lde_probe.py 20:2:1 22:2:1 22:4:2 3 > $ROOT/gpurun_out/r03_c_pmc_$m.log 2>&1
done
cd $ROOT
for m in 0 7; do echo "== F64=$m"; cat gpurun_out/r03_c_trace_$m/*/*kernel_stats.csv | grep narrow | cut -d, -f1-4; python3 tools/pmc_table.py gpurun_out/r03_c_pmc_$m narrow; done > gpurun_out/r03_c_summary.txt 2>&1
tail -5 gpurun_out/r03_c_summary.txt
