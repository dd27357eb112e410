#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (reads /root/reference/benchmarks/names.txt, which does not travel): how far is the names-like
stand-in of bench.py's `secondary.prose` from the reference's real pattern corpus?

The reference's "long" benchmark (/root/reference/benchmarks/test_comparison.py:16-31): ~4 200 lower-cased first names
of more than 4 letters over ~600-character lines of prose, a name in every 90th line.  Real names share their leading
4-grams with English words; random a-z "names-like" strings (tests/gen.py names_like) do not.  This tool compiles both
sets with the product's host compiler (no device) and simulates K1b's level 1 (tests/gen.py level1_survivor_rate: the
kernel's own pair test on the product's own table) and the exact prefix-table stage (a position is a prefix hit when
it starts with the first min(8, shortest of its group) bytes of a pattern: what the kernel hands to k_tile_main) over
the same haystack text.  Prints one line per set; the numbers go into DESIGN.md section 5."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import gen  # noqa: E402
from ahocorasick_rs_amd import capi  # noqa: E402

REF = "/root/reference/benchmarks"
LINE = ("No one who had ever seen {} in her infancy would have supposed her born to be an heroine. Her situation in life, "
        "the character of her father and mother, her own person and disposition, were all equally against her. Her father "
        "was a clergyman, without being neglected, or poor, and a very respectable man, though his name was whatevs—and "
        "he had never been handsome. He had a considerable independence besides two good livings—and he was not in the "
        "least addicted to locking up his daughters. Her mother was a woman of useful plain sense, with a good temper, and, "
        "what is more remarkable, with a good constitution {}.").lower()
# (the workload's definition: the reference's line template, test_comparison.py:22-31 -- data, as in benchmarks/bench_comparison.py)


def haystack(patterns, n_lines):
    out = []
    for i in range(n_lines):
        name = patterns[i % len(patterns)] if i % 90 == 0 else "notaperson"
        out.append(LINE.format(name, i))
    return "\n".join(out).encode("utf-8")


def prefix_hits(pats, hay: np.ndarray, q2: int) -> int:
    """positions that start with the first q2 bytes of some pattern (the set-wide Q2: an upper bound of the variable-
    length keys' hits only for groups with longer keys; names have q2 = 5)"""
    keys = {p[:q2] for p in pats}
    h = hay.tobytes()
    return sum(h.count(k) for k in keys)  # (non-overlapping count per key: keys of 5+ letters rarely self-overlap)


def main():
    with open(os.path.join(REF, "names.txt")) as f:
        real = [ln.strip().lower() for ln in f if len(ln.strip()) > 4]
    real_b = [p.encode() for p in real]
    standin = [p.encode() for p in gen.names_like(4244, 6)]
    n_lines = 8000  # ~5 MB of text
    for label, pats, hay_pats in (("real names.txt", real_b, real), ("names-like stand-in", standin, [p.decode() for p in standin])):
        hay = np.frombuffer(haystack(hay_pats, n_lines), dtype=np.uint8)
        h = capi.HostAutomaton(pats, capi.MATCH_STANDARD)
        q, q2 = int(h.t.filter_q), int(h.t.filter_q2)
        surv = gen.level1_survivor_rate(np.asarray(h.filter_xy), q, hay) if q == 5 else float("nan")
        hits = prefix_hits(pats, hay, q2)
        print(f"{label}: {len(pats)} patterns (shortest {min(map(len, pats))}), Q = {q}, Q2 = {q2}, max anchor shift {int(h.t.max_shift)}, "
              f"level-1 table density {float(h.t.filter_density):.4f}; over {len(hay) / 1e6:.1f} MB of the reference's line template: "
              f"level-1 survivors {100 * surv:.3f} % of the positions, exact prefix hits {hits} = {100 * hits / len(hay):.4f} % "
              f"({hits / (len(hay) / 4096):.2f} per 4 KiB tile)")
        h.close()


if __name__ == "__main__":
    main()
