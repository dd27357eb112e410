#!/bin/bash
# tests + the reference-scenario harness
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/misc
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison.txt 2>&1; tail -10 $OUT/bench_comparison.txt
