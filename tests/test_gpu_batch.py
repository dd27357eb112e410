"""GPU: batched and device-resident entry points, the synthetic generators,
edge cases, and a BASELINE-size (1 GiB) run checked bit-exactly."""
import random

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
KERNELS = [capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER]


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_batch_equals_singles(mk, kernel):
    rng = random.Random(50 + mk)
    pats = [bytes(rng.choice(b"abcd") for _ in range(rng.randint(1, 6))) for _ in range(40)]
    hays = [bytes(rng.choice(b"abcd") for _ in range(rng.choice([0, 0, 1, 3, 17, 200, 1500])))
            for _ in range(300)]
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    for ov in ([False, True] if mk == 0 else [False]):
        m, counts = a.find_batch(hays, overlapping=ov)
        pos = 0
        for h, c in zip(hays, counts):
            want = o.find_raw(h, overlapping=ov)
            assert int(c) == len(want)
            assert np.array_equal(cols(m[pos:pos + int(c)]), want)
            pos += int(c)
        assert pos == len(m)
    a.close()


def test_extension_batch_and_codepoints():
    import ahocorasick_rs_amd as ac
    pats = ["é☃", "ab", "b🤦", "☃"]
    hays = ["", "ab☃é☃b🤦", "xxé☃" * 50, "🤦🤦ab", "ascii only ab ab"]
    for mk in (ac.MatchKind.Standard, ac.MatchKind.LeftmostFirst, ac.MatchKind.LeftmostLongest):
        a = ac.AhoCorasick(pats, matchkind=mk)
        assert a.find_matches_as_indexes_batch(hays) == [a.find_matches_as_indexes(h) for h in hays]
    a = ac.AhoCorasick(pats)
    assert a.find_matches_as_indexes_batch(hays, overlapping=True) == \
        [a.find_matches_as_indexes(h, overlapping=True) for h in hays]
    b = ac.BytesAhoCorasick([p.encode() for p in pats])
    hb = [h.encode() for h in hays]
    assert b.find_matches_as_indexes_batch(hb) == [b.find_matches_as_indexes(h) for h in hb]
    assert b.find_matches_as_indexes_batch([]) == []


@pytest.mark.parametrize("kernel", KERNELS)
def test_device_uniform_batch_and_unaligned_pointer(kernel):
    pats = gen.gen_patterns(3000, 5, 12, gen.AZ, 1)
    n_hay, L = 512, 8192
    hay = gen.gen_textlike(n_hay * L, 13, pats)
    buf = capi.DeviceBuffer(n_hay * L + 64).upload(np.concatenate([np.zeros(5, np.uint8), hay]))
    a = capi.Automaton(pats, 0, capi.IMPL_DFA, kernel=kernel)
    o = Oracle(pats, 0, KIND_DFA)
    r = a.find_device(buf.ptr + 5, n_hay * L, n_hay=n_hay, uniform_len=L)  # ptr % 16 == 5
    m, counts = r.matches(), r.counts()
    pos = 0
    for i in range(n_hay):
        want = o.find_raw(hay[i * L:(i + 1) * L])
        assert int(counts[i]) == len(want)
        assert np.array_equal(cols(m[pos:pos + len(want)]), want)
        pos += len(want)
    r.free()
    # the same bytes as ONE haystack: matches may now span the 8 KiB cuts
    r = a.find_device(buf.ptr + 5, n_hay * L)
    assert np.array_equal(cols(r.matches()), o.find_raw(hay))
    r.free()
    a.close()


def test_generators_are_bit_exact_twins_of_numpy():
    pats = gen.gen_patterns(1000, 5, 12, gen.AZ, 1)
    a = capi.Automaton(pats, 0)
    n = 1 << 20
    buf = capi.DeviceBuffer(n)
    a.generate(buf.ptr, n, 0, 12)
    assert np.array_equal(buf.download(), gen.gen_uniform(n, gen.AZ, 12))
    a.generate(buf.ptr, n, 1, 11)
    assert np.array_equal(buf.download(), gen.gen_textlike(n, 11, pats))
    # a shard of the global stream == the same slice of the whole
    a.generate(buf.ptr, n // 2, 1, 11, stream_offset=n // 2)
    assert np.array_equal(buf.download(n // 2), gen.gen_textlike(n, 11, pats)[n // 2:])
    a.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_edge_cases(kernel):
    # match at the very start / very end, haystack shorter than the shortest pattern,
    # long patterns (chunk warm-up), all-matching haystack (occurrence buffer regrowth)
    a = capi.Automaton([b"abcde", b"cdefg"], 0, kernel=kernel)
    assert a.find_tuples(b"abcde") == [(0, 0, 5)]
    assert a.find_tuples(b"xxabcdefg") == [(0, 2, 7)]
    assert a.find_tuples(b"xxabcdefg", overlapping=True) == [(0, 2, 7), (1, 4, 9)]
    assert a.find_tuples(b"abcd") == []
    a.close()
    long_p = bytes(range(1, 200)) * 3
    a = capi.Automaton([long_p, b"\x05\x06"], 2, kernel=kernel)
    hay = b"\x00" * 777 + long_p + b"\x00" * 333 + long_p[:-1] + b"\x09" + long_p
    assert a.find_tuples(hay) == Oracle([long_p, b"\x05\x06"], 2, KIND_DFA).find(hay)
    a.close()
    a = capi.Automaton([b"aaaa", b"aa"], 0, kernel=kernel)
    hay = b"a" * 300000
    o = Oracle([b"aaaa", b"aa"], 0, KIND_DFA)
    assert np.array_equal(cols(a.find(hay)), o.find_raw(hay))
    assert np.array_equal(cols(a.find(hay, overlapping=True)), o.find_raw(hay, overlapping=True))
    a.close()


def test_baseline_size_1gib_bit_exact():
    """cfg2 at full size: 10k patterns, 1 GiB text-like haystack generated in HBM;
    the complete (pattern,start,end) stream equals the oracle's."""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    n = 1 << 30
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(n)
    a.generate(buf.ptr, n, 1, 11)
    r = a.find_device(buf.ptr, n)
    got = cols(r.matches())
    r.free()
    host = buf.download()
    want = Oracle(pats, 0, KIND_DFA).find_raw(host)
    assert len(got) == len(want) and np.array_equal(got, want)
    # size-independent property: sorted by end, non-overlapping, every slice is its pattern
    assert np.all(got[1:, 1] >= got[:-1, 2])
    for (p, s, e) in got[:: max(1, len(got) // 2000)]:
        assert host[int(s):int(e)].tobytes() == pats[int(p)]
    a.close()


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_single_haystack_cut_into_rank_ranges(mk):
    """distributed.simulate_single_sharded == every rank of find_single_sharded, run on one
    device: the ranges' matches concatenate to the whole haystack's, for every match kind."""
    from ahocorasick_rs_amd import distributed as D
    pats = gen.gen_patterns(2000, 3, 9, gen.AZ, 41) + [b"abab", b"bab", b"ababab"]
    hay = bytearray(gen.gen_textlike(1 << 20, 42, pats).tobytes())
    for cut in range(1 << 17, 1 << 20, 1 << 17):  # matches straddling the cuts of an 8-way split
        hay[cut - 5:cut + 5] = b"ababababab"
    hay = bytes(hay)
    a = capi.Automaton(pats, mk)
    for ov in ([False, True] if mk == 0 else [False]):
        whole = cols(a.find(hay, overlapping=ov))
        for world in (2, 8):
            parts = D.simulate_single_sharded(a, hay, world, overlapping=ov)
            got = np.concatenate([cols(p) for p in parts])
            assert np.array_equal(got, whole), (mk, ov, world)
    a.close()
