#!/bin/bash
# kernel-trace timelines of library variants (durations + gaps).  usage: tools/gpu_trace_ab.sh "<bench args>" name...
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/trace_ab${TAG:-}
mkdir -p $OUT
ARGS=${1:-}
shift
for v in "$@"; do
  if [ "$v" = tree ]; then unset ACX_LIB; else export ACX_LIB=/root/repo/variants/libacx_$v.so; fi
  rm -rf $OUT/$v
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -o t -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold $ARGS > $OUT/$v.log 2>&1
  cd /root/repo
  echo "== $v"; python tools/rocprof_summary.py $OUT/$v --timeline 9 | grep -v "^$" | head -30
done
