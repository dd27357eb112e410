#!/bin/bash
# round 4, call S: the PMC passes again on the final device code (a comment changed: tools/kernel_hash.py covers the text)
set -u
export TMPDIR=/tmp
cd /root/repo; mkdir -p gpurun_out/r04
timeout 1200 bash tools/collect_profiles.sh r04 pmc < /dev/null > gpurun_out/r04/collect_pmc.log 2>&1
echo "pmc rc=$?"; ls gpurun_out/r04 | grep -c pmc
