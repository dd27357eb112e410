#!/bin/bash
# round 4, call L: K0 with the automaton's table staged in LDS (small automata), same-box pair; the k_tile_main
# bucket-bound fix; new tests
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4l
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cfg1.py tests/test_api_gpu.py tests/test_gpu_parity.py -x -q > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/pytest.log
for i in 1 2; do
ACX_K0_NO_LDS_TABLE=1 timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_global_table_$i.txt 2>&1; head -9 $OUT/bench_comparison_global_table_$i.txt
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_lds_table_$i.txt 2>&1; head -9 $OUT/bench_comparison_lds_table_$i.txt
done
