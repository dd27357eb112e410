"""GPU: K0's prefilter mode (round 5, k0_small<3>) -- haystacks of up to 65 536 bytes answered by ONE launch of one
workgroup, whatever the automaton's size: level 1 on every pair of positions, the exact prefix table for the survivors,
one pattern-info load per candidate (K1b's way of finding the occurrences; the walk modes follow a chain of table gathers
from every position).  The reference's benchmark loop is calls of this size (/root/reference/benchmarks/
test_comparison.py:113-124).  Every case against the oracle; profile.small_calls says that K0 took the call."""
import random

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
SIZES = [1025, 4097, 16384, 16385, 20011, 40000, 65535, 65536]


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def sets():
    r = random.Random(3)
    az = b"abcdefghijklmnopqrstuvwxyz"
    return {
        "cfg2": gen.gen_patterns(10000, 5, 12, gen.AZ, 1),
        "names-like": [p.encode() for p in gen.names_like(4244, 6)],
        "long tails": [bytes(r.choice(az) for _ in range(r.choice([3, 7, 17, 21, 40, 100]))) for _ in range(400)],
        "anchored": list(dict.fromkeys(b"http://www." + bytes(r.choice(az) for _ in range(r.randint(4, 12))) for _ in range(2000))),
        "copies": [b"abcde", b"bcdef", b"abcde", b"cdefgh", b"abcde"] * 3,
    }


@pytest.mark.parametrize("which", ["cfg2", "names-like", "long tails", "anchored", "copies"])
def test_k0_prefilter_every_kind_and_size(which):
    pats = sets()[which]
    r = random.Random(11)
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk)
        o = Oracle(pats, mk, KIND_DFA)
        for n in SIZES + [65537]:
            hay = gen.gen_textlike(n, 31 + n, pats[:2000]).copy() if which in ("cfg2", "names-like") else \
                np.frombuffer(bytes(r.choice(b"abcdefgh ./:tpw") for _ in range(n)), dtype=np.uint8).copy()
            for _ in range(12):  # whole patterns, also flush with both ends and across the 16 KiB mark
                p = np.frombuffer(r.choice(pats), dtype=np.uint8)
                if len(p) <= n:
                    at = r.choice([0, n - len(p), max(0, min(n - len(p), 16384 - len(p) // 2)), r.randint(0, n - len(p))])
                    hay[at:at + len(p)] = p
            hay = hay.tobytes()
            for ov in ([False, True] if mk == 0 else [False]):
                want = o.find_raw(hay, overlapping=ov)
                a.profile_read(reset=True)
                got = cols(a.find(hay, overlapping=ov))
                small = a.profile_read().small_calls
                assert got.shape == want.shape and np.array_equal(got, want), (which, mk, n, ov)
                n_occ = len(Oracle(pats, 0, KIND_DFA).find_raw(hay, overlapping=True)) if which == "copies" else len(want)
                if n <= 65536 and n_occ <= 900:
                    assert small == 1, (which, mk, n, ov, small)
                if n > 65536:
                    assert small == 0
        a.close()


def test_k0_prefilter_device_pointer_alignment_and_dense_fallthrough():
    pats = gen.gen_patterns(5000, 5, 12, gen.AZ, 9)
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    hay = gen.gen_textlike(30000, 5, pats).tobytes()
    buf = capi.DeviceBuffer(len(hay) + 16)
    for lead in (0, 1, 7, 13):
        buf.upload(np.frombuffer(b"#" * lead + hay + b"#" * (16 - lead), dtype=np.uint8))
        a.profile_read(reset=True)
        r_ = a.find_device(buf.ptr + lead, len(hay))
        assert np.array_equal(cols(r_.matches()), o.find_raw(hay)), lead
        assert a.profile_read().small_calls == 1
        r_.free()
    # more than 1024 occurrences: K0 gives up, the pipeline (and its hot groups) answers
    dense = np.frombuffer(hay, dtype=np.uint8).copy()
    rng = gen.SplitMix64(1)
    for k in range(0, len(dense) - 32, 20):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        dense[k:k + len(p)] = p
    dense = dense.tobytes()
    a.profile_read(reset=True)
    assert np.array_equal(cols(a.find(dense)), o.find_raw(dense))
    assert a.profile_read().small_calls == 0
    a.close()


def test_k0_prefilter_code_points():
    spats = list(dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5)))
    bpats = [p.encode() for p in spats]
    for n in (3000, 17000, 60000):
        host = gen.gen_unicode_textlike_bytes(n, 50 + n, spats)
        cut = min(len(host), 65536)
        while (host[cut - 1] & 0xC0) == 0x80 or (cut < len(host) and (host[cut] & 0xC0) == 0x80):
            cut -= 1
        hay = host[:cut].tobytes()
        hay.decode("utf-8")
        b2c = byte_to_code_point(hay)
        for mk in (0, 2):
            a = capi.Automaton(bpats, mk)
            want = Oracle(bpats, mk, KIND_DFA).find_raw(hay)
            a.profile_read(reset=True)
            got = cols(a.find(hay, codepoints=True))
            assert a.profile_read().small_calls == 1, (n, mk)
            assert np.array_equal(got[:, 0], want[:, 0]), (n, mk)
            assert np.array_equal(got[:, 1], b2c[want[:, 1]]) and np.array_equal(got[:, 2], b2c[want[:, 2]]), (n, mk)
            a.close()


def test_k0_matches_beyond_the_result_line_are_this_calls():
    """The matches beyond the fifth travel in a pinned buffer beside the result line: two separate writes to host memory that
    nothing orders.  Eight threads on one handle, two haystacks with different matches in turn: 1 call in ~30 000 used to
    return the PREVIOUS call's entries from the sixth on (tools/stress_threads.py found it); the line now carries a hash of
    what the buffer must hold and the host reads again until it agrees."""
    import threading
    import time
    import ahocorasick_rs_amd as ac
    pats = gen.gen_patterns(3000, 4, 10, gen.AZ, 71)
    hays = [gen.gen_textlike(n, 72 + i, pats, plant_every=256).tobytes() for i, n in enumerate([5000, 40_000, 9000])]
    want = [Oracle(pats, 0, KIND_DFA).find(h) for h in hays]
    a = ac.BytesAhoCorasick(pats)
    errors, t_end = [], time.time() + 4.0

    def worker(t):
        k = t
        while time.time() < t_end and not errors:
            i = k % len(hays)
            if a.find_matches_as_indexes(hays[i]) != want[i]:
                errors.append((t, i))
            k += 1

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:3]
