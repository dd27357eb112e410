#!/bin/bash
# round 4, call P: K0 kernel durations by input (rocprofv3 kernel trace, the .db files come back in gpurun_out)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4p
mkdir -p $OUT
cd /root/repo
for ds in short short_nomatch short_onematch long; do
  timeout 60 python tools/k0_probe.py $ds indexes 2000 < /dev/null
  ( cd /tmp && timeout 120 rocprofv3 --kernel-trace -d $OUT/prof_$ds -o k0 -- python /root/repo/tools/k0_probe.py $ds indexes 2000 < /dev/null ) > $OUT/prof_$ds.log 2>&1
  echo "prof $ds rc=$?"
done
