#!/bin/bash
# quick per-kernel numbers of the headline and the large-set / str configurations (no tests)
set -u
export TMPDIR=/tmp
cd /root/repo
tools/gpu_ablate.sh cfg2 none 2>&1 | grep -v "^\[" | head -5
tools/gpu_ablate.sh cfg4 none 2>&1 | head -4
tools/gpu_ablate.sh cfg5 none 2>&1 | head -5
