#!/bin/bash
# round 4, call O: K0 writes its records as one coalesced run (LDS image) -- tests, then the reference's benchmark loop
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4o
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_cfg1.py tests/test_api_gpu.py tests/test_gpu_parity.py -x -q < /dev/null > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -3 $OUT/pytest.log < /dev/null
timeout 200 python benchmarks/bench_comparison.py < /dev/null > $OUT/bench_comparison.txt 2>&1; head -9 $OUT/bench_comparison.txt < /dev/null
timeout 60 python tools/k0_probe.py short indexes 2000 < /dev/null
timeout 60 python tools/k0_probe.py long indexes 2000 < /dev/null
