#!/bin/bash
# round 4: randomised parity on the final kernels (anchors, side test, tile-ordered dense path), three seeds
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04_fuzz
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "count_exchange" > $OUT/pytest_rccl.log 2>&1
echo "rccl c-abi test (own process) rc=$?"; tail -3 $OUT/pytest_rccl.log
for seed in 40401 40402 40403; do
  timeout 400 python tools/gpu_fuzz.py 150 $seed > $OUT/fuzz_$seed.log 2>&1
  echo "fuzz seed $seed rc=$?"; tail -2 $OUT/fuzz_$seed.log
done
