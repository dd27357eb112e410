#!/bin/bash
# round 3 GPU helper: [tests] the round-3 tests  [bench] bench lines (default with cold + 8 GiB, U, dense)  [suite] whole GPU suite
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r3a
mkdir -p $OUT
cd /root/repo
WHAT=${1:-tests}
if [ "$WHAT" = tests ]; then
  timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_fuzz.py -x -q -m gpu --durations=15 > $OUT/pytest_new.log 2>&1
  echo "pytest new rc=$?"; tail -40 $OUT/pytest_new.log
fi
if [ "$WHAT" = suite ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/pytest_all.log 2>&1
  echo "pytest all rc=$?"; tail -45 $OUT/pytest_all.log
fi
if [ "$WHAT" = bench ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
  cut -c1-2500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
  for d in U D; do
    timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dist $d > $OUT/bench_$d.json 2> $OUT/bench_$d.err
    cut -c1-1200 $OUT/bench_$d.json; tail -2 $OUT/bench_$d.err
  done
  ACX_NO_BUCKET=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target-size > $OUT/bench_T_dense_path.json 2> $OUT/bench_T_dense_path.err
  cut -c1-1200 $OUT/bench_T_dense_path.json; tail -2 $OUT/bench_T_dense_path.err
fi
