#!/bin/bash
# Round 5, VERDICT item 3: VALU wave-instructions of a WHOLE proof, per proof workload of bench.py, from the kernels that actually
# run in this build: one prover (throughput profile, as bench.py's provers) under
#   rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU     (PMC only, its own run per workload)
# summarised per kernel into gpurun_out/r05_pmc_proofs.json together with the SHA-256 of the libp3hip.so that ran; bench.py's
# valu_roofline reads that file from profiles/ and marks its figures stale when the loaded library is another build.
set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
#            key          hash      log_n hiding proofs blowup
for spec in "cfg2:poseidon2:20:0:6:1" "cfg2_keccak:keccak:20:0:6:1" "cfg2_keccak_hiding:keccak:20:1:6:1" "cfg3:poseidon2:24:0:2:2"; do
  IFS=: read key hash logn hid n blow <<< "$spec"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/gpurun_out/r05_pmc_proof_$key -- python3 $ROOT/tools/prove_n.py $hash $logn $hid $n throughput $blow > $ROOT/gpurun_out/r05_pmc_proof_$key.log 2>&1
  echo "$key done" >> $ROOT/gpurun_out/r05_pmc_proofs_progress.txt
done
cd $ROOT
python3 tools/r05_pmc_proofs_summarize.py
