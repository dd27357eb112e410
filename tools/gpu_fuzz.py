#!/usr/bin/env python3
"""Randomised parity run on a GPU box: HIP path (C ABI) vs the oracle on random pattern sets
(alphabets from 2 letters to all bytes to UTF-8 with 2/3/4-byte characters, 1 .. 30 000
patterns, duplicates, nested patterns) and haystacks of 0 .. 48 MiB (text-like, uniform, dense,
planted), all match kinds, overlapping, byte offsets and code points, both scan kernels.
usage: gpu_fuzz.py [seconds] [seed]   -- prints one line per case, exits non-zero on a mismatch.
tests/test_gpu_fuzz.py runs a bounded, seeded slice of it (fuzz(budget, seed, max_size_log2)) under -m gpu.
UTF-8 cases only ever hold patterns made of whole characters (a str pattern cannot end inside one:
the precondition of codepoints = 1, include/acx.h)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point
from ahocorasick_rs_amd import capi

rng = random.Random(20260927)
MAX_SIZE_LOG2 = 25.5
ALPHAS = [("ab", b"ab"), ("abc", b"abc"), ("a-h", b"abcdefgh"), ("a-z", gen.AZ), ("a-z+sp", gen.AZ + b" "),
          ("bytes", bytes(range(256))), ("utf8", None)]
UNI = "abcdefghijklmnopqrstuvwxyz" + "é☃\U0001F926"


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def make_case():
    name, alpha = ALPHAS[rng.randrange(len(ALPHAS))]
    n_pat = int(10 ** rng.uniform(0, 4.5))
    lo = rng.choice([1, 2, 3, 5, 5, 5, 8])
    hi = lo + rng.choice([0, 3, 7, 20, 60])
    if alpha is None:
        pats = ["".join(rng.choice(UNI) for _ in range(rng.randint(lo, hi))).encode() for _ in range(n_pat)]
    else:
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(lo, hi))) for _ in range(n_pat)]
    if rng.random() < 0.3 and n_pat > 3:  # duplicates and nested patterns
        pats += [pats[rng.randrange(n_pat)] for _ in range(n_pat // 10 + 1)]
        if alpha is None:  # (prefixes of whole characters: a str pattern cannot end inside one)
            pats += [pats[rng.randrange(n_pat)].decode()[:rng.randint(1, 4)].encode() for _ in range(n_pat // 10 + 1)]
        else:
            pats += [pats[rng.randrange(n_pat)][:rng.randint(1, 6)] for _ in range(n_pat // 10 + 1)]
    size = int(2 ** rng.uniform(0, MAX_SIZE_LOG2)) if rng.random() < 0.9 else rng.choice([0, 1, 4095, 4096, 4097, 262144, 262145])
    if alpha is not None and len(alpha) <= 8:  # small alphabets: the output is many times the input -- keep it bounded
        size = min(size, (1 << 18) if lo <= 2 else (1 << 21))
    kind = rng.choice(["uniform", "text", "planted", "dense"])
    if alpha is None:
        chars = [rng.choice(UNI) if rng.random() < 0.3 else rng.choice("abcdefgh ") for _ in range(min(size, 200000))]
        base = "".join(chars).encode()
        hay = (base * (size // max(len(base), 1) + 1))[:size]
        while hay and (hay[-1] & 0xC0) == 0x80 or (hay and hay[-1] >= 0xC0):
            hay = hay[:-1]  # cut at a character boundary
        hay = bytearray(hay)
    else:
        a = np.frombuffer(alpha, dtype=np.uint8)
        hay = bytearray(a[np.random.default_rng(rng.randrange(1 << 30)).integers(0, len(a), size)].tobytes())
    if kind in ("planted", "dense") and size > 64:
        step = rng.choice([37, 300, 5000]) if kind == "planted" else rng.choice([3, 9, 17])
        p = 0
        while p < size - 70 and (kind == "planted" or p < 400000):
            x = pats[rng.randrange(len(pats))]
            if alpha is not None and p + len(x) <= size:
                hay[p:p + len(x)] = x
            p += step + rng.randrange(step)
    return name, kind, pats, bytes(hay)



def fuzz(budget: float, seed: int, max_size_log2: float = 25.5, save_failures: bool = True):
    """-> (cases, failures); deterministic in (seed, max_size_log2) up to where the time budget cuts it"""
    global rng, MAX_SIZE_LOG2
    rng = random.Random(seed)
    MAX_SIZE_LOG2 = max_size_log2
    t_end = time.time() + budget
    cases = fails = 0
    while time.time() < t_end:
        name, kind, pats, hay = make_case()
        mk = rng.randrange(3)
        kernel = rng.choice([None, capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER])
        ov = mk == 0 and rng.random() < 0.4
        cp = name == "utf8" and rng.random() < 0.7
        if os.environ.get("FUZZ_ONLY") and os.environ["FUZZ_ONLY"] != name:
            continue
        # hundreds of copies of every pattern on text where every position matches (seed 40404, case 23: 22 264 patterns over
        # {a, b} of 1-4 bytes = 30 distinct strings, 256 KiB of a/b).  The device reports one occurrence per STRING and the
        # copies are expanded where the result is complete ("copies of a pattern"; tests/test_gpu_copies.py),
        # so such a case costs its OUTPUT only -- which for an overlapping search is the occurrences x the copies: skipped
        # when that is beyond the 2 * 10^7 rows the rule below allows (decided here, before the oracle builds them; the rng
        # stream is not disturbed)
        copies = len(pats) / max(len(set(pats)), 1)
        if ov and copies > 50 and len(hay) * copies > 2e7:
            print(f"skip {name:7s} {kind:8s} pats {len(pats):6d} hay {len(hay):9d}: overlapping, {copies:.0f} copies of every pattern", flush=True)
            continue
        try:
            a = capi.Automaton(pats, mk, kernel=kernel)
        except capi.AcxError as e:
            print("build error", e); continue
        o = Oracle(pats, mk, KIND_DFA)
        want = o.find_raw(hay, overlapping=ov)
        if cp and len(want):
            b2c = byte_to_code_point(hay)
            want = np.stack([want[:, 0], b2c[want[:, 1].astype(np.int64)], b2c[want[:, 2].astype(np.int64)]], 1)
        if len(want) > 20_000_000:
            a.close(); continue
        try:
            got = cols(a.find(hay, overlapping=ov, codepoints=cp))
        except (capi.AcxError, ValueError, MemoryError) as e:
            print("find error", name, kind, len(pats), len(hay), mk, ov, cp, kernel, len(want), e, flush=True)
            fails += 1; a.close(); continue
        ok = np.array_equal(got, want)
        # the same bytes at an unaligned device address, and again (buffers of the previous call in flight)
        got2 = cols(a.find(np.frombuffer(b"x" * 3 + hay, dtype=np.uint8)[3:], overlapping=ov, codepoints=cp))
        ok = ok and np.array_equal(got2, want)
        cases += 1
        if not ok:
            for tag, g in (("aligned", got), ("unaligned", got2)):
                if np.array_equal(g, want):
                    continue
                m = min(len(g), len(want))
                d = np.nonzero((g[:m] != want[:m]).any(1))[0]
                i = int(d[0]) if len(d) else m
                print(f"  {tag}: got {len(g)} want {len(want)} rows, first difference at row {i}: got "
                      f"{g[i].tolist() if i < len(g) else None} want {want[i].tolist() if i < len(want) else None}", flush=True)
        print(f"{'ok  ' if ok else 'FAIL'} {name:7s} {kind:8s} pats {len(pats):6d} hay {len(hay):9d} mk {mk} ov {int(ov)} cp {int(cp)} "
              f"kernel {kernel} matches {len(want)}", flush=True)
        if not ok:
            fails += 1
            if save_failures:
                np.save(f"/root/repo/gpurun_out/fuzz_fail_{cases}_hay.npy", np.frombuffer(hay, dtype=np.uint8))
                open(f"/root/repo/gpurun_out/fuzz_fail_{cases}_pats.txt", "w").write(repr((pats, mk, ov, cp, kernel)))
        a.close()
    return cases, fails


if __name__ == "__main__":
    n_cases, n_fails = fuzz(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0,
                            int(sys.argv[2]) if len(sys.argv) > 2 else 20260927)
    print(f"{n_cases} cases, {n_fails} failures")
    sys.exit(1 if n_fails else 0)
