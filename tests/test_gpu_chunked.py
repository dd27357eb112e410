"""GPU: a call that would enumerate more occurrences than one pass can index is cut into byte ranges (round 5).

The reference's loop is an iterator: it has no limit on what it reports (/root/reference/src/lib.rs:53, 59).  One pass of
the device pipeline indexes its occurrences with 32 bits; until round 5 a call beyond that raised ACX_ETOOBIG.  Now the
haystack is searched in byte ranges, one after the other -- overlapping: a range reports what ENDS in it; non-overlapping:
a range reports what STARTS in it, and the iteration resumes where the last match of the ranges in front ended -- and the
pieces are spliced with global offsets (acx_api.cpp, run_chunked).  2^32 occurrences need > 100 GB of matches: the tests
lower the limit of one pass (ACX_MAX_OCC) or force the cut (ACX_CHUNK_BYTES), both read per call; acx_path_stats counts the
ranges.  Everything is compared with the oracle, element-wise."""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def long_patterns(seed: int):
    # short and long patterns with common material: matches that span a range border, borders inside a match, nested ones
    import random
    r = random.Random(seed)
    base = [bytes(r.choice(b"abcd") for _ in range(r.randint(2, 6))) for _ in range(60)]
    longs = [b"".join(r.choice(base) for _ in range(r.randint(4, 12))) for _ in range(40)]
    return list(dict.fromkeys(base + longs))


@pytest.mark.parametrize("piece", [997, 4096, 50_000])
def test_forced_ranges_every_kind(monkeypatch, piece):
    pats = long_patterns(3)
    m = max(len(p) for p in pats)
    assert piece > 2 * m + 16
    import random
    r = random.Random(piece)
    hay = bytes(r.choice(b"abcd") for _ in range(150_000))
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk)
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            monkeypatch.setenv("ACX_CHUNK_BYTES", str(piece))
            a.path_stats(reset=True)
            got = cols(a.find(hay, overlapping=ov))
            st = a.path_stats()
            monkeypatch.delenv("ACX_CHUNK_BYTES")
            assert st["byte_ranges"] >= len(hay) // piece, st
            assert got.shape == want.shape, (mk, ov, got.shape, want.shape)
            assert np.array_equal(got, want), (mk, ov)
        a.close()


def test_forced_ranges_on_the_pipeline_and_with_code_points(monkeypatch):
    # ranges long enough for the general pipeline (K1b, hit slots, the tile kernels), a device-resident haystack at an odd
    # address, and the str API: the pieces run on byte offsets, the code points are taken once over the whole result
    pats = gen.gen_patterns(3000, 5, 12, gen.AZ, 8)
    hay = gen.gen_textlike((3 << 20) + 4321, 21, pats).tobytes()
    for mk in (0, 2):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk, capi.IMPL_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            monkeypatch.setenv("ACX_CHUNK_BYTES", str(700_001))
            a.path_stats(reset=True)
            got = cols(a.find(hay, overlapping=ov))
            st = a.path_stats()
            assert st["byte_ranges"] == 5 and st["k0"] == 0, st
            assert np.array_equal(got, want), (mk, ov)
            buf = capi.DeviceBuffer(len(hay) + 3)
            buf.upload(np.frombuffer(b"xyz" + hay, dtype=np.uint8))
            r = a.find_device(buf.ptr + 3, len(hay), overlapping=ov)  # (device-resident, odd address: the ranges' too)
            got = cols(r.matches())
            r.free()
            buf.free()
            monkeypatch.delenv("ACX_CHUNK_BYTES")
            assert np.array_equal(got, want), (mk, ov, "device")
        a.close()
    spats = list(dict.fromkeys(gen.gen_patterns(2000, 3, 9, gen.AZ_UNI, 5)))
    bpats = [p.encode() for p in spats]
    hay = gen.gen_unicode_textlike_bytes(1 << 20, 56, spats).tobytes()
    hay.decode("utf-8")
    b2c = byte_to_code_point(hay)
    for mk in (0, 1):
        a = capi.Automaton(bpats, mk)
        want = Oracle(bpats, mk, KIND_DFA).find_raw(hay)
        monkeypatch.setenv("ACX_CHUNK_BYTES", str(300_007))  # (ranges begin inside characters)
        got = cols(a.find(hay, codepoints=True))
        monkeypatch.delenv("ACX_CHUNK_BYTES")
        assert np.array_equal(got[:, 0], want[:, 0])
        assert np.array_equal(got[:, 1], b2c[want[:, 1]]) and np.array_equal(got[:, 2], b2c[want[:, 2]])
        a.close()


def test_a_pass_over_its_limit_is_cut_until_the_pieces_fit(monkeypatch):
    # the real trigger, with the limit of one pass lowered: a haystack with a pattern every 32 bytes enumerates ~130 000
    # occurrences on the dense path -- halved until every piece stays under 20 000, no error, oracle-identical
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = gen.gen_uniform(4 << 20, gen.AZ, 12).copy()
    rng = gen.SplitMix64(77)
    for k in range(0, len(hay) - 32, 32):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p
    hay = hay.tobytes()
    for mk, ov in ((0, False), (0, True), (2, False)):
        a = capi.Automaton(pats, mk, capi.IMPL_DFA)
        want = Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping=ov)
        # both forms of the dense path count their occurrences against the limit, and so does the hot pipeline when it
        # sizes the output of a call whose groups are (nearly) all hot -- the way this input goes when nothing is forced
        for way in ("dense_tiles", "dense_radix", "hot"):
            monkeypatch.setenv("ACX_MAX_OCC", "20000")
            if way != "hot":
                monkeypatch.setenv("ACX_NO_BUCKET", "1")
            if way == "dense_radix":
                monkeypatch.setenv("ACX_NO_DENSE_TILES", "1")
            a.path_stats(reset=True)
            got = cols(a.find(hay, overlapping=ov))
            st = a.path_stats()
            monkeypatch.delenv("ACX_MAX_OCC")
            monkeypatch.delenv("ACX_NO_BUCKET", raising=False)
            monkeypatch.delenv("ACX_NO_DENSE_TILES", raising=False)
            assert st["byte_ranges"] >= (2 if way == "hot" else 8) and (way == "hot" or st[way] >= 8), (way, st)
            assert got.shape == want.shape and np.array_equal(got, want), (mk, ov, way)
        a.close()


def test_a_batch_over_the_limit_is_cut_at_haystack_boundaries(monkeypatch):
    # round 6: a batch whose occurrences one pass cannot index is searched in two parts, cut at a haystack boundary (each
    # part again a batch, a part of one haystack a call of its own that may go on in byte ranges); until then: ACX_ETOOBIG
    pats = [b"ab", b"b", b"bab"]
    import random
    r = random.Random(5)
    hays = [b"ab" * r.randint(1, 60_000) + bytes(r.choice(b"abc") for _ in range(r.randint(0, 300))) for _ in range(7)] + [b"", b"abab"]
    for mk, ov in ((0, False), (0, True), (1, False), (2, False)):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk)
        want = [o.find_raw(h, overlapping=ov) for h in hays]
        monkeypatch.setenv("ACX_MAX_OCC", "50000")
        monkeypatch.setenv("ACX_NO_BUCKET", "1")
        a.path_stats(reset=True)
        m, counts = a.find_batch(hays, overlapping=ov)
        st = a.path_stats()
        monkeypatch.delenv("ACX_MAX_OCC")
        monkeypatch.delenv("ACX_NO_BUCKET")
        assert st["byte_ranges"] >= 2, st
        assert list(counts) == [len(x) for x in want], (mk, ov)
        got = cols(m)
        assert np.array_equal(got, np.concatenate([x.reshape(-1, 3) for x in want]).astype(np.uint64)), (mk, ov)
        a.close()


def test_a_uniform_batch_and_code_points_over_the_limit(monkeypatch):
    pats = ["é☃", "☃", "a☃é", "ab"]
    a = capi.Automaton([p.encode() for p in pats], 2)
    o = Oracle([p.encode() for p in pats], 2, KIND_DFA)
    text = ("aé☃☃ab" * 4000)
    hays = [text.encode()] * 9
    monkeypatch.setenv("ACX_MAX_OCC", "20000")
    monkeypatch.setenv("ACX_NO_BUCKET", "1")
    m, counts = a.find_batch(hays, codepoints=True)
    one = o.find_raw(hays[0])
    b2c = byte_to_code_point(hays[0])
    want_one = np.stack([one[:, 0], b2c[one[:, 1]], b2c[one[:, 2]]], 1).astype(np.uint64)
    assert list(counts) == [len(one)] * 9
    assert np.array_equal(cols(m), np.concatenate([want_one] * 9))
    # ... and haystacks of one length, resident on the device (byte offsets)
    blob = np.frombuffer(b"".join(hays), dtype=np.uint8)
    buf = capi.DeviceBuffer(blob.size).upload(blob)
    res = a.find_device(buf.ptr, blob.size, n_hay=9, uniform_len=len(hays[0]))
    monkeypatch.delenv("ACX_MAX_OCC")
    monkeypatch.delenv("ACX_NO_BUCKET")
    got = cols(res.matches())
    assert np.array_equal(got, np.concatenate([one.astype(np.uint64)] * 9))
    assert list(res.counts()) == [len(one)] * 9
    res.free()
    buf.free()
    a.close()
