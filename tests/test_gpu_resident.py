"""GPU: the RESIDENT K0 (round 6, kernels.hip k0_resident) -- a loop of calls on short host haystacks (the reference's own
benchmark: /root/reference/benchmarks/test_comparison.py:113-124, one call per 75-byte haystack) is answered by one
workgroup that stays on the device between the calls and is fed through a mailbox in pinned host memory; a call costs a
poll on either side instead of a launch.  Every case against the oracle; path_stats says which way the calls went
(k0 = calls K0 answered, resident_launches = launches they cost)."""
import os
import random
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")

FEW = [b"abc", b"hello", b"b", b"aardvark", b"fish", b"whatwhat", b"sixteen-bytes-xy", b"ninebytes", b"host7", b"host76", b"b"]


def sets():
    """One automaton per way K0 finds the occurrences (kernels.hip, k0_call<MODE>)."""
    return {
        "direct comparison": FEW,                                          # MODE 2
        "table in LDS": FEW + [b"seventeen-bytes-xy"],                     # MODE 1
        "tables in global memory": FEW + gen.gen_patterns(3000, 4, 12, gen.AZ, 3),  # MODE 0, MODE 3 beyond 1 KiB
    }


def tuples(arr):
    return [(int(p), int(s), int(e)) for (p, s, e) in arr]


def haystacks(pats, n, seed, lo=1, hi=400):
    r = random.Random(seed)
    words = [p for p in pats[:200]] + [b"xyz", b" ", b"qq", b"the", b"\xc3\xa9", b"\xf0\x9f\xa4\xa6"]
    out = []
    for _ in range(n):
        want = r.randint(lo, hi)
        h = b""
        while len(h) < want:
            h += r.choice(words) + (b" " if r.random() < 0.5 else b"")
        out.append(h[:want])
    return out


@pytest.mark.parametrize("which", ["direct comparison", "table in LDS", "tables in global memory"])
def test_resident_loop_every_mode_every_kind_bytes_and_code_points(which):
    pats = sets()[which]
    hays = haystacks(pats, 300, 5) + haystacks(pats, 40, 6, 1025, 5000) + [b"b", b"x", b"b" * 7, b"hello fish abc b host7 host76"]
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk)
        for ov in ([False, True] if mk == 0 else [False]):
            for cp in (False, True):
                want = [o.find(h, overlapping=ov) for h in hays]
                if cp:  # (src/lib.rs:73-88, oracle/ac_oracle.c aco_byte_to_code_point)
                    maps = [byte_to_code_point(h) for h in hays]
                    want = [[(p, int(m[s]), int(m[e])) for p, s, e in w] for w, m in zip(want, maps)]
                a.path_stats(reset=True)
                got = [tuples(a.find(h, overlapping=ov, codepoints=cp)) for h in hays]  # (the loop: nothing between the calls)
                st = a.path_stats()
                assert got == want, (which, mk, ov, cp, next(i for i in range(len(hays)) if got[i] != want[i]))
                dense = sum(1 for w_ in want if len(w_) > 900)
                assert st["k0"] >= len(hays) - dense, (which, mk, ov, cp, st)
                assert 1 <= st["resident_launches"], st
        a.close()


def test_resident_code_points_exact():
    """The str API's indexes through the resident kernel: two- and four-byte characters in front of and between the matches."""
    import ahocorasick_rs_amd as ac
    spats = [p.decode() for p in FEW] + ["é", "🤦b"]
    bpats = [p.encode() for p in spats]
    r = random.Random(9)
    pieces = ["é", "🤦b", " hello ", "☃", " fish ", "abc", "🤦", "host76", "ééé", " b", "whatwhat", "x", "日本"]
    texts = ["".join(r.choice(pieces) for _ in range(r.randint(1, 40))) for _ in range(300)]
    kinds = [(ac.MatchKind.Standard, 0), (ac.MatchKind.LeftmostFirst, 1), (ac.MatchKind.LeftmostLongest, 2)]
    for mkind, mk in kinds:
        a = ac.AhoCorasick(spats, matchkind=mkind)
        o = Oracle(bpats, mk, KIND_DFA)
        wants = []
        for t in texts:
            bts = t.encode()
            cp = np.cumsum(np.frombuffer(bts, dtype=np.uint8) & 0xC0 != 0x80) - 1
            cp = np.concatenate([cp, [cp[-1] + 1]])
            wants.append([(int(p), int(cp[s]), int(cp[e])) for p, s, e in o.find_raw(bts)])
        got = [a.find_matches_as_indexes(t) for t in texts]
        assert got == wants, mk
        assert [a.find_matches_as_strings(t) for t in texts] == [[spats[p] for p, _, _ in w] for w in wants]


def test_resident_launches_are_few_and_an_idle_kernel_leaves():
    pats = sets()["table in LDS"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    hays = haystacks(pats, 2000, 7, 20, 120)
    want = [o.find(h) for h in hays]
    a.find(hays[0])
    a.path_stats(reset=True)
    got = [tuples(a.find(h)) for h in hays]
    st = a.path_stats()
    assert got == want
    assert st["k0"] == len(hays)
    # (a kernel lives for at most a millisecond and leaves after 200 us without a call: a loop of ~10 us calls costs a
    # launch per ~100 calls; a slow or disturbed host more -- but not one per call)
    assert st["resident_launches"] <= len(hays) // 4, st
    # idle: the kernel has left when the next call comes -- that call is the first of a new launch
    for k in range(5):
        time.sleep(0.01)
        a.path_stats(reset=True)
        assert tuples(a.find(hays[k])) == want[k]
        assert a.path_stats()["resident_launches"] == 1
    a.close()


def test_resident_gives_way_to_every_other_call_of_its_context():
    """A pipeline call, a device-memory call, a batch, a dense small call (more than 1 024 occurrences: the pipeline
    answers) between the loop's calls: the kernel is told to leave first (streams may share a hardware queue), the next
    small call launches the next one."""
    pats = sets()["tables in global memory"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    small = haystacks(pats, 60, 8, 30, 300)
    big = gen.gen_textlike(1 << 20, 3, pats).tobytes()
    dense = b"b" * 3000
    buf = capi.DeviceBuffer(len(big))
    buf.upload(np.frombuffer(big, dtype=np.uint8))
    want_small = [o.find(h) for h in small]
    want_big = o.find_raw(big)
    want_dense = o.find(dense)
    for rep in range(3):
        for k, h in enumerate(small):
            assert tuples(a.find(h)) == want_small[k]
            if k % 20 == 5:
                got = a.find(big)
                assert np.array_equal(np.stack([got["pattern"], got["start"], got["end"]], 1), want_big)
            if k % 20 == 10:
                r_ = a.find_device(buf.ptr, len(big))
                m = r_.matches()
                assert np.array_equal(np.stack([m["pattern"], m["start"], m["end"]], 1), want_big)
                r_.free()
            if k % 20 == 15:
                assert tuples(a.find(dense)) == want_dense
            if k % 20 == 18:
                got, counts = a.find_batch(small[:7])
                assert [int(c) for c in counts] == [len(w) for w in want_small[:7]]
    a.close()


def test_resident_alternating_kinds_of_call_fall_back_to_plain_launches():
    pats = sets()["direct comparison"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    hays = haystacks(pats, 400, 10, 10, 90)
    want = [o.find(h, overlapping=bool(k & 1)) for k, h in enumerate(hays)]
    a.path_stats(reset=True)
    for k, h in enumerate(hays):
        assert tuples(a.find(h, overlapping=bool(k & 1))) == want[k], k
    st = a.path_stats()
    assert st["k0"] == len(hays)
    assert st["resident_launches"] < 40, st  # (four changes of kind, then 256 calls as plain launches)
    a.close()


def test_resident_from_threads_one_automaton():
    pats = sets()["tables in global memory"]
    a = capi.Automaton(pats, 1)
    o = Oracle(pats, 1, KIND_DFA)
    hays = haystacks(pats, 500, 12, 5, 900)
    want = [o.find(h) for h in hays]
    bad = []

    def worker(t):
        for rep in range(4):
            for k in range(len(hays)):
                i = (k * 7 + t) % len(hays)
                if tuples(a.find(hays[i])) != want[i]:
                    bad.append((t, i))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad[:5]
    a.close()


SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join(os.environ["ACX_ROOT"], "tests")); sys.path.insert(0, os.environ["ACX_ROOT"])
import gen
from oracle_lib import KIND_DFA, Oracle
from ahocorasick_rs_amd import capi
import test_gpu_resident as T
pats = T.sets()[sys.argv[1]]
hays = T.haystacks(pats, 3000, 13, 5, 600)
for mk in (0, 2):
    a = capi.Automaton(pats, mk)
    o = Oracle(pats, mk, KIND_DFA)
    want = [o.find(h) for h in hays]
    a.path_stats(reset=True)
    got = [T.tuples(a.find(h)) for h in hays]
    assert got == want, next(i for i in range(len(hays)) if got[i] != want[i])
    st = a.path_stats()
    print("STATS", mk, st["k0"], st["resident_launches"])
    a.close()
print("OK")
"""


@pytest.mark.parametrize("env,which", [({"ACX_RESIDENT_LIFE_US": "30"}, "table in LDS"),
                                        ({"ACX_RESIDENT_LIFE_US": "30"}, "tables in global memory"),
                                        ({"ACX_RESIDENT_IDLE_US": "3"}, "direct comparison"),
                                        ({"ACX_NO_RESIDENT": "1"}, "table in LDS")])
def test_resident_short_lives_and_switched_off(env, which):
    """A life of 30 us / an idle limit of 3 us: the kernel leaves between (and under) the calls all the time -- every call
    still gets its answer, from the kernel that took it or from the next launch.  ACX_NO_RESIDENT=1: plain launches only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", SCRIPT, which], env={**os.environ, **env, "ACX_ROOT": root}, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    stats = [l.split() for l in r.stdout.splitlines() if l.startswith("STATS")]
    for _, mk, k0, launches in stats:
        assert int(k0) == 3000
        if "ACX_NO_RESIDENT" in env:
            assert int(launches) == 0
        else:
            assert int(launches) > 30, stats  # (they do leave)
