#!/usr/bin/env python3
"""Per-call latency and batch throughput of the drop-in API on the reference's own
benchmark scenarios (reference: benchmarks/test_comparison.py:16-166 -- "short": 10 patterns x
10 000 haystacks of ~75 characters; "long": ~4 200 name-like patterns x 100 000 haystacks of
~600 characters, 1 in 90 containing one pattern; standard / indexes / overlapping / longest
match, each as a Python loop of single-haystack calls).

On a GPU a single short haystack is launch-latency bound, so two shapes are timed:
  loop   the reference's shape: one find_matches_as_* call per haystack
  batch  the same haystacks through find_matches_as_indexes_batch (one device pass)
and the results of the two are checked against each other.  No pyahocorasick, no
pytest-benchmark (neither is in the image); prints one table.

usage: python benchmarks/bench_comparison.py [--loop-haystacks N] [--names FILE]
"""
import argparse
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


import gen  # noqa: E402  (tests/gen.py: the seeded inputs)
from oracle_lib import KIND_DFA, Oracle  # noqa: E402  (the checker; TEST INFRASTRUCTURE, timed beside the product)


def datasets(names_file):
    short_p = ["abc", "hello", "world", "aardvark", "fish", "what", "arbitrarymonkey", "birds", "host7", "host76"]
    short_h = ["arbitrarymonkey says hello to fish host76, 0.123 my friend, but why??? {}".format(i)
               for i in range(10_000)]
    if names_file and os.path.exists(names_file):
        long_p = [l.strip().lower() for l in open(names_file) if len(l.strip()) > 4]
    else:
        long_p = gen.names_like()  # seeded stand-in
    long_h = gen.names_lines(long_p, 100_000, every=90)
    return {"short": (short_p, short_h), "long": (long_p, long_h)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loop-haystacks", type=int, default=2000, help="haystacks timed in the per-call loop")
    ap.add_argument("--names", default="/root/reference/benchmarks/names.txt")
    args = ap.parse_args()
    import ahocorasick_rs_amd as ac

    rows = []
    for ds, (pats, hays) in datasets(args.names).items():
        sub = hays[:args.loop_haystacks]
        nbytes_all = sum(len(h) for h in hays)
        scenarios = [
            ("standard strings", dict(), "find_matches_as_strings", dict()),
            ("standard indexes", dict(), "find_matches_as_indexes", dict()),
            ("overlapping strings", dict(), "find_matches_as_strings", dict(overlapping=True)),
            ("longest match strings", dict(matchkind=ac.MatchKind.LeftmostLongest), "find_matches_as_strings", dict()),
        ]
        for label, ckw, meth, kw in scenarios:
            t0 = time.perf_counter()
            a = ac.AhoCorasick(pats, **ckw)
            build_ms = (time.perf_counter() - t0) * 1e3
            f = getattr(a, meth)
            f(sub[0], **kw)  # warm-up (first launch, workspace allocation)
            # (the results are kept for the check below: lists of tuples are containers, and the collector's passes over
            # everything alive -- the 100 000 haystacks among it -- were 3-4 us per call of the "indexes" loops, more than
            # half of what a call costs since round 6; the reference's harness keeps no results.  Off for the timed loop.)
            gc.collect()
            gc.disable()
            t0 = time.perf_counter()
            loop_out = [f(h, **kw) for h in sub]
            loop_us = (time.perf_counter() - t0) / len(sub) * 1e6
            gc.enable()
            a.find_matches_as_indexes_batch(hays[:100], **kw)
            t0 = time.perf_counter()
            batch_out = a.find_matches_as_indexes_batch(hays, **kw)
            batch_s = time.perf_counter() - t0
            # the batch is the loop, one device pass: same answers
            for h, lo, bo in zip(sub, loop_out, batch_out):
                want = [h[s:e] for _, s, e in bo] if meth.endswith("strings") else bo
                assert lo == want, (ds, label)
            # the CPU beside it: the oracle (C restatement of the reference's algorithm) called once per
            # haystack through ctypes, the shape of the reference's own loop.  The checker, timed -- never
            # a path of the product.
            mk = 2 if "longest" in label else 0
            o = Oracle([x.encode() for x in pats], mk, KIND_DFA)
            enc = [h.encode() for h in sub]
            ov = bool(kw.get("overlapping"))
            o.find_raw(enc[0], overlapping=ov)
            t0 = time.perf_counter()
            for h in enc:
                o.find_raw(h, overlapping=ov)
            cpu_us = (time.perf_counter() - t0) / len(enc) * 1e6
            rows.append((ds, label, len(pats), build_ms, loop_us, batch_s / len(hays) * 1e6,
                         nbytes_all / batch_s / 1e6, sum(len(m) for m in batch_out), cpu_us))
    print(f"{'dataset':8s} {'scenario':24s} {'patterns':>8s} {'build ms':>9s} {'loop us/hay':>12s} "
          f"{'batch us/hay':>13s} {'batch MB/s':>11s} {'matches':>9s} {'CPU oracle us/hay':>18s}")
    for r in rows:
        print(f"{r[0]:8s} {r[1]:24s} {r[2]:8d} {r[3]:9.1f} {r[4]:12.1f} {r[5]:13.3f} {r[6]:11.1f} {r[7]:9d} {r[8]:18.2f}")
    crossover()


def crossover():
    """Where one call on the GPU overtakes one call on a CPU core: the long-dataset automaton over a
    single haystack of growing size (bytes API, byte offsets), GPU = BytesAhoCorasick call (K0 up to
    16 KiB, the general pipeline beyond; host memory in, Python list out), CPU = the oracle's DFA
    loop on one core."""
    import ahocorasick_rs_amd as ac
    names = gen.names_like()
    pats = [x.encode() for x in names]
    a = ac.BytesAhoCorasick(pats)
    o = Oracle(pats, 0, KIND_DFA)
    print("\ncrossover: one call, the long-dataset automaton (4 244 patterns), haystack size growing")
    print(f"{'bytes':>10s} {'GPU us/call':>12s} {'CPU us/call':>12s} {'GPU MB/s':>10s} {'CPU MB/s':>10s}")
    for n in (64, 1024, 16384, 65536, 1 << 20, 16 << 20, 256 << 20):
        hay = gen.names_haystack(names, n, every=3)
        reps = max(3, min(2000, (64 << 20) // n))
        a.find_matches_as_indexes(hay)
        t0 = time.perf_counter()
        for _ in range(reps):
            g = a.find_matches_as_indexes(hay)
        gpu = (time.perf_counter() - t0) / reps
        creps = max(1, min(reps, (16 << 20) // n))
        o.find_raw(hay)
        t0 = time.perf_counter()
        for _ in range(creps):
            c = o.find_raw(hay)
        cpu = (time.perf_counter() - t0) / creps
        assert [tuple(int(v) for v in r) for r in c] == g
        print(f"{n:10d} {gpu * 1e6:12.1f} {cpu * 1e6:12.1f} {n / gpu / 1e6:10.1f} {n / cpu / 1e6:10.1f}")


if __name__ == "__main__":
    main()
