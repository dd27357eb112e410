#!/bin/bash
# round 4, call N: where a K0 call's microseconds are -- kernel durations by dataset (rocprofv3), host loop beside them
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4n
mkdir -p $OUT
cd /root/repo
for ds in short long; do
  python tools/k0_probe.py $ds indexes 2000
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_$ds -o k0 -- python /root/repo/tools/k0_probe.py $ds indexes 2000 ) > $OUT/prof_$ds.log 2>&1
  tail -2 $OUT/prof_$ds.log
  f=$(find $OUT/prof_$ds -name '*kernel_stats.csv' | head -1); head -5 $f
done
