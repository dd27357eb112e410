"""GPU: copies of a pattern in an OVERLAPPING search cost their records, nothing else (round 5).

The reference keeps every copy of a string and an overlapping search reports them all, together, ids ascending
(/root/reference/src/lib.rs:53-54 -> try_find_overlapping_iter; one state's match list in the order the patterns were added).
Until round 5 the device enumerated, verified, staged and sorted every copy's occurrence: hundreds of copies of every
string on text where every position matches sent every call to the radix-sort form (and against the 2^32 limit of one
pass).  Now the search runs on the view without the later copies -- one occurrence per string, under the lowest id, the
view non-overlapping searches take since round 4 -- and the complete result is expanded (acx_api.cpp expand_copies: a
prefix sum over the copies' counts, one thread per output record; K0's pinned result on the host; a batch's per-haystack
counts follow).  Every case against the oracle, element-wise, through every entry point."""
import random

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def many_copies(copies: int, seed: int):
    r = random.Random(seed)
    strings = list(dict.fromkeys(bytes(r.choice(b"ab") for _ in range(r.randint(1, 4))) for _ in range(200)))
    pats = [s for s in strings for _ in range(copies)]
    r.shuffle(pats)  # (the copies' ids interleave)
    return pats


@pytest.mark.parametrize("kernel", [None, "dfa_walk", "prefilter"])
def test_hundreds_of_copies_where_every_position_matches(kernel):
    pats = many_copies(150, 4)
    o = Oracle(pats, 0, KIND_DFA)
    k = {None: None, "dfa_walk": capi.KERNEL_DFA_WALK, "prefilter": capi.KERNEL_PREFILTER}[kernel]
    a = capi.Automaton(pats, 0, kernel=k)
    r = random.Random(9)
    for n in (1, 37, 300, 5000, 20_000):  # K0 (its result expanded on the host) ... the pipeline
        hay = bytes(r.choice(b"ab") for _ in range(n))
        want = o.find_raw(hay, overlapping=True)
        got = cols(a.find(hay, overlapping=True))
        assert got.shape == want.shape, (n, got.shape, want.shape)
        assert np.array_equal(got, want), n
        # device-resident haystack (K0 and the pipeline write device memory: expanded there)
        buf = capi.DeviceBuffer(n + 1)
        buf.upload(np.frombuffer(b"x" + hay, dtype=np.uint8))
        res = a.find_device(buf.ptr + 1, n, overlapping=True)
        got = cols(res.matches())
        res.free(); buf.free()
        assert np.array_equal(got, want), (n, "device")
        # (the non-overlapping search of the same handle: the lowest ids only, as before)
        assert np.array_equal(cols(a.find(hay)), o.find_raw(hay)), n
    a.close()


@pytest.mark.parametrize("force", [None, "1"])
def test_a_few_copies_in_a_large_set_sparse_path_batch_and_code_points(monkeypatch, force):
    # force None: a set with a few copies enumerates them on the device, as before (the expansion is a pass over the result and
    # a round trip: taken when a quarter of the ids are copies); "1" (ACX_EXPAND_COPIES, read when the automaton is built):
    # the expansion all the same -- both ways are the reference's answer
    if force:
        monkeypatch.setenv("ACX_EXPAND_COPIES", force)
    base = gen.gen_patterns(3000, 4, 10, gen.AZ, 5)
    r = random.Random(2)
    pats = list(base)
    for _ in range(400):  # some strings twice, some five times, anywhere in the id order
        s = r.choice(base)
        for _ in range(r.choice((1, 1, 4))):
            pats.insert(r.randrange(len(pats) + 1), s)
    o = Oracle(pats, 0, KIND_DFA)
    a = capi.Automaton(pats, 0)
    hay = gen.gen_textlike(2 << 20, 31, base).tobytes()
    want = o.find_raw(hay, overlapping=True)
    a.path_stats(reset=True)
    got = cols(a.find(hay, overlapping=True))
    st = a.path_stats()
    assert st["sparse"] == 1 and st["dense_radix"] == 0, st
    assert np.array_equal(got, want)
    assert len(want) > len(o.find_raw(hay))  # (copies were reported)
    # in byte ranges (the pieces are expanded, then spliced)
    monkeypatch.setenv("ACX_CHUNK_BYTES", "300001")
    assert np.array_equal(cols(a.find(hay, overlapping=True)), want)
    monkeypatch.delenv("ACX_CHUNK_BYTES")
    # batch: local offsets, per-haystack counts of the expanded result
    hays = [gen.gen_textlike(20_000 + 13 * i, 40 + i, base).tobytes() for i in range(50)] + [b"", b"q"]
    m, counts = a.find_batch(hays, overlapping=True)
    at = 0
    for i, h in enumerate(hays):
        w = o.find_raw(h, overlapping=True)
        assert counts[i] == len(w), i
        assert np.array_equal(cols(m[at:at + len(w)]), w), i
        at += len(w)
    assert at == len(m)
    a.close()
    # str API: code points of the expanded records
    spats = gen.gen_patterns(500, 2, 6, gen.AZ_UNI, 5)
    spats = spats + spats[::3] + spats[::7]
    bp = [p.encode() for p in spats]
    hay = gen.gen_unicode_textlike_bytes(300_000, 56, spats).tobytes()
    b2c = byte_to_code_point(hay)
    a = capi.Automaton(bp, 0)
    want = Oracle(bp, 0, KIND_DFA).find_raw(hay, overlapping=True)
    got = cols(a.find(hay, overlapping=True, codepoints=True))
    assert np.array_equal(got[:, 0], want[:, 0])
    assert np.array_equal(got[:, 1], b2c[want[:, 1]]) and np.array_equal(got[:, 2], b2c[want[:, 2]])
    a.close()


def test_the_facade_reports_every_copy():
    # through the package (the C++ CPython extension over the same C ABI): the reference's own API shape
    import ahocorasick_rs_amd as ac
    pats = ["ab", "b", "ab", "xab", "b", "ab"]
    a = ac.AhoCorasick(pats)
    assert a.find_matches_as_indexes("xab", overlapping=True) == [(3, 0, 3), (0, 1, 3), (2, 1, 3), (5, 1, 3), (1, 2, 3), (4, 2, 3)]
    assert a.find_matches_as_strings("xab", overlapping=True) == ["xab", "ab", "ab", "ab", "b", "b"]
