#!/bin/bash
# per-kernel times of a configuration under measurement switches: tools/gpu_ablate.sh cfg5 "mk=ll,cp=0" "mk=standard,cp=0" ...
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ablate
mkdir -p $OUT
CFG=$1; shift
for ab in "$@"; do
  tag=$(echo "${CFG}_$ab" | tr '=,' '__')
  cd /tmp; rm -rf $OUT/$tag
  if [ "$ab" = none ]; then A=""; else A="--ablate $ab"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --config $CFG $A > $OUT/$tag.log 2>&1
  cd /root/repo
  echo "== $CFG $ab: $(grep -o '"value": [0-9.]*' $OUT/$tag.log | head -1) $(grep -o '"matches_total": [0-9]*' $OUT/$tag.log) $(grep -o '"prefix_hits_per_step": [0-9]*' $OUT/$tag.log)"
  python tools/rocprof_summary.py $OUT/$tag 2>/dev/null | head -8 | tail -7
done
