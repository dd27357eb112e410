#!/bin/bash
# one-off measurements: new GPU test, reference-scenario harness, batch workload bench, PCIe-inclusive rate
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/misc
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "single_haystack" 2>&1 | tail -2
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison.txt 2>&1; cat $OUT/bench_comparison.txt | tail -12
timeout 200 python bench.py --steps 20 --warmup 3 --workload batch --no-cpu-baseline > $OUT/bench_batch.json 2> $OUT/bench_batch.err; cut -c1-260 $OUT/bench_batch.json
timeout 300 python - > $OUT/pcie.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gen
from ahocorasick_rs_amd import capi
pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
a = capi.Automaton(pats, 0, capi.IMPL_DFA)
n = 1 << 30
hay = gen.gen_textlike(n, 11, pats) if False else None
buf = capi.DeviceBuffer(n); a.generate(buf.ptr, n, 1, 11); hay = buf.download()
for rep in range(3):
    t0 = time.perf_counter(); m = a.find(hay); dt = time.perf_counter() - t0
    print(f"acx_find host->host 1 GiB pageable: {dt*1e3:.1f} ms  {n/dt/1e9:.2f} GB/s  matches {len(m)}")
PY
cat $OUT/pcie.txt
