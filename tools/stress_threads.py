#!/usr/bin/env python3
"""Stress form of tests/test_api_gpu.py::test_concurrent_searches_from_threads: N threads on ONE automaton (4 contexts), haystacks
of every path (K0 walk / prefilter, the pipeline, hot groups), for a number of seconds; prints every mismatch with what differs.
usage: stress_threads.py [seconds] [threads] [small]
small: every thread loops over SHORT haystacks of one kind of call -- the resident K0's case (round 6: a workgroup per context that
stays on the device and is fed through pinned host memory); the calls' count is what the run is for (10^6 and more)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import gen
from oracle_lib import KIND_DFA, Oracle
from ahocorasick_rs_amd import BytesAhoCorasick

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
small = len(sys.argv) > 3 and sys.argv[3] == "small"
pats = gen.gen_patterns(3000, 4, 10, gen.AZ, 71)
sizes = [300, 75, 17, 1000, 640, 33, 1, 900, 2000, 12, 480, 5000] if small else [300, 5000, 70_000, 1 << 20, 3 << 20, 17, 0, 40_000]
hays = [gen.gen_textlike(n, 72 + i, pats, plant_every=64 if small else 256).tobytes() for i, n in enumerate(sizes)]
o = Oracle(pats, 0, KIND_DFA)
want = [o.find(h) for h in hays]
want_ov = [o.find(h, overlapping=True) for h in hays]
shared = BytesAhoCorasick(pats)
errors, calls = [], [0]
t_end = time.time() + secs


def worker(t):
    rep = 0
    while time.time() < t_end:
        for k in range(len(hays)):
            i = (k + t + rep) % len(hays)
            for ov, expect in (((bool(t & 1), (want_ov if t & 1 else want)[i]),) if small else ((False, want[i]), (True, want_ov[i]))):
                got = shared.find_matches_as_indexes(hays[i], overlapping=ov)
                calls[0] += 1
                if got != expect:
                    d = next((q for q, (x, y) in enumerate(zip(got, expect)) if x != y), min(len(got), len(expect)))
                    errors.append((ov, t, i, len(hays[i]), len(got), len(expect), d, got[max(0, d - 1):d + 2], expect[max(0, d - 1):d + 2]))
        rep += 1


ts = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"{calls[0]} calls, {len(errors)} mismatches" + (f", paths {shared.path_stats()}" if hasattr(shared, "path_stats") else ""))
for e in errors[:20]:
    print(e)
sys.exit(1 if errors else 0)
