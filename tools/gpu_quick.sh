#!/bin/bash
# quick GPU check: tests, bench T/U, kernel trace.  Every command bounded by timeout.
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/quick
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 200 python bench.py --steps 20 --warmup 3 > $OUT/bench_T.json 2> $OUT/bench_T.err; cut -c1-300 $OUT/bench_T.json; grep -o '"roofline".*' $OUT/bench_T.json | cut -c1-300
timeout 200 python bench.py --steps 20 --warmup 3 --dist U --no-cpu-baseline > $OUT/bench_U.json 2> $OUT/bench_U.err; cut -c1-200 $OUT/bench_U.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_T.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $OUT/trace_T 2>/dev/null | head -30
