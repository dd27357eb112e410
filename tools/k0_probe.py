#!/usr/bin/env python3
"""K0 per call, by dataset (benchmarks/bench_comparison.py's "short" / "long"), for rocprofv3 --kernel-trace --stats:
which part of a call's microseconds is the kernel's.  usage: k0_probe.py short|long|short_nomatch|short_onematch [indexes|strings] [calls]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
import bench_comparison as bc  # noqa: E402
import ahocorasick_rs_amd as ac  # noqa: E402

ds = sys.argv[1]
meth = "find_matches_as_" + (sys.argv[2] if len(sys.argv) > 2 else "indexes")
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
base = ds.split("_")[0]
pats, hays = bc.datasets("/nonexistent")[base]
if ds.endswith("_nomatch"):    # the same haystacks, a set of the same shape that never occurs
    pats = ["zqabc", "zqhello", "zqworld", "zqaardvark", "zqfish", "zqwhat", "zqarbitrarymonkey", "zqbirds", "zqhost7", "zqhost76"]
elif ds.endswith("_onematch"): # one short match per haystack
    pats = ["zqabc", "zqhello", "zqworld", "zqaardvark", "fish", "zqwhat", "zqarbitrarymonkey", "zqbirds", "zqhost7", "zqhost76"]
a = ac.AhoCorasick(pats)
f = getattr(a, meth)
sub = hays[:calls]
for h in sub[:200]:
    f(h)
t0 = time.perf_counter()
n = 0
for h in sub:
    n += len(f(h))
dt = time.perf_counter() - t0
print(f"{ds} {meth}: {dt / len(sub) * 1e6:.2f} us per call, {n} matches in {len(sub)} calls, haystack bytes avg {sum(map(len, sub)) / len(sub):.0f}")
