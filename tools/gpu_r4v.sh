#!/bin/bash
# round 4, call V: seed 40404 again with the generator's skip rule (it stopped at case 23 before), then one more seed
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04
mkdir -p $OUT
cd /root/repo
timeout 110 python tools/gpu_fuzz.py 60 40404 < /dev/null > $OUT/fuzz_40404_with_skip.log 2>&1
echo "fuzz 40404 rc=$?"; tail -2 $OUT/fuzz_40404_with_skip.log < /dev/null; grep -c "^skip" $OUT/fuzz_40404_with_skip.log < /dev/null
timeout 80 python tools/gpu_fuzz.py 45 40406 < /dev/null > $OUT/fuzz_40406.log 2>&1
echo "fuzz 40406 rc=$?"; tail -1 $OUT/fuzz_40406.log < /dev/null
