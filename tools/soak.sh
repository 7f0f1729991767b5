#!/bin/bash
# tools/soak.sh <seconds> <seed0> [nproc]: the randomised parity soak (tools/fuzz_parity.py) in `nproc` processes with
# consecutive seeds, while tests/csrc/hammer_case scores the saved round-3 case (tests/golden/fuzz_31337.npz) again and
# again on the same GPU from a second process with its own co-running 200 k-point load -- timing perturbation for both.
# Everything lands in gpurun_out/soak_*.log; exit code 1 if any process reported a mismatch.
secs=${1:-600}; seed=${2:-1000}; np=${3:-4}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R; mkdir -p gpurun_out
python tools/case_blob.py tests/golden/fuzz_31337.npz gpurun_out/fuzz_31337.blob > /dev/null
timeout -s KILL $((secs+60)) tests/csrc/hammer_case gpurun_out/fuzz_31337.blob $secs device load > gpurun_out/soak_hammer_$seed.log 2>&1 &
hp=$!
pids=""
for k in $(seq 0 $((np-1))); do
  # (--trace: the parameters of every case before its call; timeout: a process that hangs is killed a minute after its budget --
  #  its log then ends with the case it hung in, and the soak reports it instead of running into gpurun's limit)
  OMP_NUM_THREADS=1 timeout -s KILL $((secs+60)) python tools/fuzz_parity.py $secs $((seed+k)) --trace --log gpurun_out/soak_rng_$((seed+k)).jsonl > gpurun_out/soak_fuzz_$((seed+k)).log 2>&1 &
  pids="$pids $!"
done
rc=0
for p in $pids; do wait $p || rc=1; done
wait $hp || rc=1
for f in gpurun_out/soak_fuzz_*.log; do grep -q "^fuzz parity:\|MISMATCH" $f || echo "HUNG OR KILLED: $f, last case: $(grep '^case' $f | tail -n 1)"; done
grep -h "^fuzz parity:\|MISMATCH" gpurun_out/soak_fuzz_*.log; tail -n 1 gpurun_out/soak_hammer_$seed.log
exit $rc
