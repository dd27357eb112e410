#!/bin/bash
# round 4: the whole GPU suite on the final tree, then the round's evidence (tools/collect_profiles.sh r04)
set -u
export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r04
cd /root/repo
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04/pytest_gpu.log 2>&1
echo "gpu suite rc=$?"; tail -4 gpurun_out/r04/pytest_gpu.log
timeout 2400 bash tools/collect_profiles.sh r04 all > gpurun_out/r04/collect.log 2>&1
echo "collect rc=$?"; tail -5 gpurun_out/r04/collect.log
