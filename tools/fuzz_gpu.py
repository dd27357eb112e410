#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison across the execution paths (K0, K1a / K1b + tile
kernels, region mode, batches).  Not part of the test suites (it is open-ended); every
failure prints the seed that reproduces it.  usage: python tools/fuzz_gpu.py [seconds] [seed]"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import KIND_DFA, Oracle  # noqa: E402
from ahocorasick_rs_amd import capi  # noqa: E402


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def one(seed):
    rng = random.Random(seed)
    alpha = rng.choice([b"ab", b"abc", b"abcdefgh", bytes(range(97, 123)), bytes(range(256))])
    npat = rng.choice([1, 3, 20, 200, 3000])
    lo, hi = rng.choice([(1, 3), (2, 6), (3, 9), (5, 12), (4, 40)])
    pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(lo, hi))) for _ in range(npat)]
    n = rng.choice([5, 300, 5000, 16384, 16385, 70000, 262144 + 17, 700000, 1 << 20])
    noise = np.random.default_rng(seed).integers(0, len(alpha), n)
    hay = np.frombuffer(alpha, dtype=np.uint8)[noise].copy()
    for _ in range(rng.randint(0, 200)):  # plant patterns, clustered around tile / bucket borders
        p = rng.choice(pats)
        if len(p) > n:
            continue
        border = rng.choice([0, 4096, 8192, 262144, 524288, n]) + rng.randint(-40, 40)
        at = max(0, min(n - len(p), rng.choice([border, rng.randint(0, n - len(p))])))
        hay[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
    hay = hay.tobytes()
    mk = rng.randint(0, 2)
    kernel = rng.choice([None, None, capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER])
    try:
        a = capi.Automaton(pats, mk, kernel=kernel)
    except capi.AcxError:
        a = capi.Automaton(pats, mk)  # prefilter not available for this pattern set
    o = Oracle(pats, mk, KIND_DFA)
    for ov in ([False, True] if mk == 0 else [False]):
        for rep in range(2):  # twice: the second call runs on the state the first one left
            got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
            if not np.array_equal(got, want):
                raise AssertionError(f"seed {seed}: mk {mk} ov {ov} kernel {kernel} n {n} got {len(got)} want {len(want)}")
    if n <= 70000:  # the same bytes as a ragged batch
        cuts = sorted(set([0, n] + [rng.randint(0, n) for _ in range(rng.randint(0, 30))]))
        hays = [hay[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
        m, counts = a.find_batch(hays)
        pos = 0
        for h, c in zip(hays, counts):
            want = o.find_raw(h)
            if int(c) != len(want) or not np.array_equal(cols(m[pos:pos + int(c)]), want):
                raise AssertionError(f"seed {seed}: batch mismatch mk {mk} kernel {kernel}")
            pos += int(c)
    a.close()


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, k = time.time(), 0
    while time.time() - t0 < budget:
        one(seed + k)
        k += 1
    print(f"fuzz: {k} cases from seed {seed} OK in {time.time() - t0:.0f} s")
