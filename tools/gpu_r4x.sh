#!/bin/bash
# round 4, call X: the copy view under the seeded fuzz slice, the batch and the API tests
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4x
mkdir -p $OUT
cd /root/repo
ACX_FUZZ_SECONDS=30 timeout 85 python -m pytest tests/test_gpu_fuzz.py tests/test_api_gpu.py tests/test_gpu_batch.py -x -q < /dev/null > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/pytest.log < /dev/null
