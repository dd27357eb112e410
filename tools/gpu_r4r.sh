#!/bin/bash
# round 4, call R: K0 direct comparison for a handful of short patterns -- tests, the reference's benchmark loop
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4r
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_cfg1.py tests/test_api_gpu.py tests/test_gpu_parity.py -x -q < /dev/null > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log < /dev/null
timeout 200 python benchmarks/bench_comparison.py < /dev/null > $OUT/bench_comparison.txt 2>&1; head -9 $OUT/bench_comparison.txt < /dev/null
for ds in short short_nomatch short_onematch long; do timeout 60 python tools/k0_probe.py $ds indexes 2000 < /dev/null; done
ACX_K0_NO_DIRECT=1 timeout 60 python tools/k0_probe.py short indexes 2000 < /dev/null
