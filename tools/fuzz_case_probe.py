#!/usr/bin/env python3
"""Replays one case of tools/gpu_fuzz.py (seed, case number) at growing haystack sizes and prints the call times:
usage: fuzz_case_probe.py seed case_no kernel(-1 = default) size...   (each size in its own bounded run)"""
import os, random, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)

if sys.argv[1] == "--one":
    import numpy as np
    import gpu_fuzz as F
    from ahocorasick_rs_amd import capi
    seed, case_no, kernel, size = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    F.rng = random.Random(seed); F.MAX_SIZE_LOG2 = 25.5
    for _ in range(case_no):
        name, kind, pats, hay = F.make_case()
        mk = F.rng.randrange(3); F.rng.choice([None, 1, 2]); ov = mk == 0 and F.rng.random() < 0.4; cp = name == "utf8" and F.rng.random() < 0.7
    hay = hay[:size]
    a = capi.Automaton(pats, mk, kernel=None if kernel < 0 else kernel)
    for rep in range(2):
        t = time.time(); got = a.find(hay, overlapping=ov); dt = time.time() - t
        print(f"size {len(hay)} kernel {kernel} mk {mk} ov {ov}: {dt * 1e3:.1f} ms, {len(got)} matches, stats {a.last_stats() if hasattr(a, 'last_stats') else ''}", flush=True)
    a.close()
else:
    seed, case_no, kernel = sys.argv[1:4]
    for size in sys.argv[4:]:
        try:
            subprocess.run([sys.executable, __file__, "--one", seed, case_no, kernel, size], timeout=45, stdin=subprocess.DEVNULL)
        except subprocess.TimeoutExpired:
            print(f"size {size} kernel {kernel}: more than 45 s", flush=True)
