#!/bin/bash
# round 4, call M: K0 for tiny haystacks (wave scans, slice-wise lead counts, class window in registers)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4m
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cfg1.py tests/test_api_gpu.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2; do
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_$i.txt 2>&1; head -9 $OUT/bench_comparison_$i.txt
ACX_K0_NO_LDS_TABLE=1 timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_global_table_$i.txt 2>&1; head -9 $OUT/bench_comparison_global_table_$i.txt
done
