#!/bin/bash
# bench T/U + kernel trace only (no tests)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/quick
mkdir -p $OUT
cd /root/repo
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_T.json 2> $OUT/bench_T.err; cut -c1-200 $OUT/bench_T.json
timeout 200 python bench.py --steps 20 --warmup 3 --dist U --no-cpu-baseline > $OUT/bench_U.json 2> $OUT/bench_U.err; cut -c1-200 $OUT/bench_U.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_T.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $OUT/trace_T 2>/dev/null | grep -E "walk|k1b|tile|fill|copy"
