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
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def datasets(names_file):
    import gen
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
            t0 = time.perf_counter()
            loop_out = [f(h, **kw) for h in sub]
            loop_us = (time.perf_counter() - t0) / len(sub) * 1e6
            a.find_matches_as_indexes_batch(hays[:100], **kw)
            t0 = time.perf_counter()
            batch_out = a.find_matches_as_indexes_batch(hays, **kw)
            batch_s = time.perf_counter() - t0
            # the batch is the loop, one device pass: same answers
            for h, lo, bo in zip(sub, loop_out, batch_out):
                want = [h[s:e] for _, s, e in bo] if meth.endswith("strings") else bo
                assert lo == want, (ds, label)
            rows.append((ds, label, len(pats), build_ms, loop_us, batch_s / len(hays) * 1e6,
                         nbytes_all / batch_s / 1e6, sum(len(m) for m in batch_out)))
    print(f"{'dataset':8s} {'scenario':24s} {'patterns':>8s} {'build ms':>9s} {'loop us/hay':>12s} "
          f"{'batch us/hay':>13s} {'batch MB/s':>11s} {'matches':>9s}")
    for r in rows:
        print(f"{r[0]:8s} {r[1]:24s} {r[2]:8d} {r[3]:9.1f} {r[4]:12.1f} {r[5]:13.3f} {r[6]:11.1f} {r[7]:9d}")


if __name__ == "__main__":
    main()
