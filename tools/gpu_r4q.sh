#!/bin/bash
# round 4, call Q: K0's polled result as one 64-byte line (no fence round trips) -- tests, the reference's benchmark loop
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4q
mkdir -p $OUT
cd /root/repo
timeout 400 python -m pytest tests/test_gpu_cfg1.py tests/test_api_gpu.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q < /dev/null > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log < /dev/null
timeout 200 python benchmarks/bench_comparison.py < /dev/null > $OUT/bench_comparison.txt 2>&1; head -9 $OUT/bench_comparison.txt < /dev/null
for ds in short short_nomatch short_onematch long; do timeout 60 python tools/k0_probe.py $ds indexes 2000 < /dev/null; done
ACX_SMALL_SYNC=1 timeout 60 python tools/k0_probe.py short indexes 2000 < /dev/null
