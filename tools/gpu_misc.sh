#!/bin/bash
# kernel trace of the cfg3-shape (batch) workload at N = 1
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/misc
rm -rf $OUT/batchtrace; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/batchtrace -o t -- python /root/repo/bench.py --steps 10 --warmup 3 --workload batch --no-cpu-baseline > $OUT/batchtrace.log 2>&1
python /root/repo/tools/rocprof_summary.py $OUT/batchtrace --timeline 12 2>/dev/null | tail -32
cd /root/repo; timeout 200 python bench.py --steps 20 --warmup 3 --workload batch --no-cpu-baseline | cut -c1-180
