#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity for them lives in
tests/test_gpu_configs.py; this only times them), device-resident haystacks:
  cfg4  100 000 patterns (a-z / bytes), overlapping=True, iid-uniform haystack
  cfg5  10 000 patterns over a-z + 2/3/4-byte UTF-8 characters, LeftmostLongest, code-point indexes
usage: python tools/measure_configs.py [MiB]   (default 256)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen  # noqa: E402
from ahocorasick_rs_amd import capi  # noqa: E402

MIB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = MIB << 20


def timed(a, ptr, n, **kw):
    for _ in range(2):
        a.find_device(ptr, n, **kw).free()
    a.profile_enable(True)
    a.profile_read(reset=True)
    a.find_device(ptr, n, **kw).free()
    pr = a.profile_read(reset=True)
    a.profile_enable(False)
    print(f"  [profile] scan {pr.scan_ms:.3f} ms  post {pr.post_ms:.3f} ms  prefix hits {pr.prefix_hits}  "
          f"occurrences {pr.raw_occurrences}", flush=True)
    t0 = time.perf_counter()
    steps = 5
    for _ in range(steps):
        r = a.find_device(ptr, n, **kw)
        cnt = r.count
        r.free()
    return (time.perf_counter() - t0) / steps, cnt


rows = []
for label, alpha, seed in (("cfg4 a-z", gen.AZ, 3), ("cfg4 bytes", gen.ALL_BYTES, 4)):
    pats = gen.gen_patterns(100000, 5, 12, alpha, seed)
    t0 = time.perf_counter()
    a = capi.Automaton(pats, 0, capi.IMPL_AUTO)
    build_s = time.perf_counter() - t0
    hay = gen.gen_uniform(N, alpha, 12)
    buf = capi.DeviceBuffer(N)
    buf.upload(hay)
    dt, cnt = timed(a, buf.ptr, N, overlapping=True)
    i = a.info
    rows.append((label + " overlapping", i.n_states, i.table_bytes / 2**20, capi.KERNEL_NAMES[i.kernel], build_s,
                 dt * 1e3, N / dt / 1e9, cnt))
    a.close()

pats = list(dict.fromkeys(gen.gen_patterns(10000, 5, 12, gen.AZ_UNI, 5)))
hay = gen.gen_unicode_textlike(N // 2, 56, pats).encode()  # ~N/2 code points
t0 = time.perf_counter()
a = capi.Automaton([p.encode() for p in pats], 2, capi.IMPL_AUTO)
build_s = time.perf_counter() - t0
buf = capi.DeviceBuffer(len(hay))
buf.upload(np.frombuffer(hay, dtype=np.uint8))
dt, cnt = timed(a, buf.ptr, len(hay), codepoints=True)
i = a.info
rows.append(("cfg5 utf-8 leftmost-longest cp", i.n_states, i.table_bytes / 2**20, capi.KERNEL_NAMES[i.kernel],
             build_s, dt * 1e3, len(hay) / dt / 1e9, cnt))
a.close()

print(f"{'config':34s} {'states':>8s} {'DFA MiB':>8s} {'kernel':>10s} {'build s':>8s} {'ms/pass':>8s} {'GB/s':>8s} {'matches':>9s}")
for r in rows:
    print(f"{r[0]:34s} {r[1]:8d} {r[2]:8.1f} {r[3]:>10s} {r[4]:8.2f} {r[5]:8.3f} {r[6]:8.1f} {r[7]:9d}")
