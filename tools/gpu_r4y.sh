#!/bin/bash
# round 4, call Y: the copy view through the DFA walk kernels too (host tables only)
set -u
export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r4y
cd /root/repo
timeout 38 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cfg1.py -x -q -k "copies or cfg1" < /dev/null > gpurun_out/r4y/pytest.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r4y/pytest.log < /dev/null
