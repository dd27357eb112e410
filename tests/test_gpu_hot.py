"""GPU: the HOT pipeline (round 5) -- a dense stretch of the input costs the groups it lies in, not the call.

The reference's loop costs the same per byte wherever the matches are (/root/reference/src/lib.rs:53, 59).  Until round
5 ONE 4 KiB tile with more than 64 prefix hits sent the whole call to the dense path and kept the handle there for eight
more calls.  Now K1b files the hits beyond a tile's slots in an overflow list, k_tile_main lists the groups it cannot
finish, the sparse kernels finish every other group, the listed groups go through the tile-ordered dense machinery
(k_hot_verify -> k_dense_main) and k_tile_write splices both by the groups' counts.  Every case is compared with the
oracle, element-wise, and the handle's path counters (acx_path_stats) say which way the call went."""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle, byte_to_code_point

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
GROUP = 64 * 4096  # bytes of index space per group of k_tile_main


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def plant(hay: np.ndarray, pats, lo: int, hi: int, every: int, seed: int) -> None:
    rng = gen.SplitMix64(seed)
    for k in range(lo, hi - 32, every):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p


def check_all_kinds(pats, hay: bytes, expect_hot: bool, impl=None):
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk, capi.IMPL_DFA if impl is None else impl)
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            a.path_stats(reset=True)
            got = cols(a.find(hay, overlapping=ov))
            want = o.find_raw(hay, overlapping=ov)
            assert got.shape == want.shape, (mk, ov, got.shape, want.shape)
            assert np.array_equal(got, want), (mk, ov)
            st = a.path_stats()
            if expect_hot:
                assert st["hot_calls"] == 1 and st["dense_tiles"] == 0 and st["dense_radix"] == 0, (mk, ov, st)
        a.close()


def test_one_hot_region_costs_its_groups_not_the_call():
    # the headline's set over 4 MiB of text with ONE 64 KiB region that holds a pattern every 32 bytes (bench.py --dist H1)
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = gen.gen_textlike(4 << 20, 11, pats).copy()
    a0 = (len(hay) // 3) & ~(GROUP - 1)
    plant(hay, pats, a0 + 8 * 4096, a0 + 8 * 4096 + (64 << 10), 32, 77)
    check_all_kinds(pats, hay.tobytes(), True)
    # the sparse path takes the next call of the same handle again: nothing is held on the dense path
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    o = Oracle(pats, 0, KIND_DFA)
    assert np.array_equal(cols(a.find(hay.tobytes())), o.find_raw(hay.tobytes()))
    plain = gen.gen_textlike(4 << 20, 12, pats).tobytes()
    a.path_stats(reset=True)
    assert np.array_equal(cols(a.find(plain)), o.find_raw(plain))
    st = a.path_stats()
    assert st["sparse"] == 1 and st["hot_calls"] == 0 and st["dense_tiles"] == 0 and st["dense_radix"] == 0, st
    a.close()


@pytest.mark.parametrize("where", ["first_tiles", "last_tiles", "across_groups", "context_only", "every_other_group"])
def test_hot_regions_at_the_seams(where):
    # dense stretches laid over the places where the hot pipeline and the sparse kernels meet: the stream's first and
    # last tiles, a group boundary (both groups hot, the second one's context is the first one's tiles), the last tiles
    # of a group only (they are the NEXT group's context: it must go hot with them), alternating hot and sparse groups
    pats = gen.gen_patterns(3000, 5, 12, gen.AZ, 8)
    n = 6 * GROUP + 12345
    hay = gen.gen_textlike(n, 21, pats).copy()
    if where == "first_tiles":
        plant(hay, pats, 0, 3 * 4096, 32, 1)
    elif where == "last_tiles":
        plant(hay, pats, n - 3 * 4096, n, 32, 2)
    elif where == "across_groups":
        plant(hay, pats, 2 * GROUP - 5 * 4096, 2 * GROUP + 5 * 4096, 32, 3)
    elif where == "context_only":
        plant(hay, pats, 3 * GROUP - 2 * 4096, 3 * GROUP - 100, 32, 4)
    else:
        for g in (0, 2, 4):
            plant(hay, pats, g * GROUP + 4096, g * GROUP + 40 * 4096, 48, 5 + g)
    check_all_kinds(pats, hay.tobytes(), True)
    check_all_kinds(pats, hay.tobytes()[5:], True)  # (an unaligned view: 16-byte lead, other tile borders)


def test_bucket_and_group_overflow_stay_local():
    # not the slots but the stage of k_tile_main overflows: more than 24 occurrences in ONE 4 KiB bucket of an otherwise
    # sparse stream, and more than 1024 reported matches in one group without a full bucket -- both used to redo the call
    pats = [b"abcde", b"cde"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    hay = bytearray(b"." * (1 << 20))
    hay[50000:50000 + 5 * 40] = b"abcde" * 40
    for p in range(1000, len(hay) - 10, 3001):
        hay[p:p + 5] = b"abcde"
    a.path_stats(reset=True)
    for ov in (False, True):
        assert np.array_equal(cols(a.find(bytes(hay), overlapping=ov)), o.find_raw(bytes(hay), overlapping=ov))
    st = a.path_stats()
    assert st["hot_calls"] == 2 and st["hot_groups"] <= 4 and st["dense_tiles"] == st["dense_radix"] == 0, st
    hay = bytearray(b"." * (3 * GROUP))
    for p in range(GROUP, 2 * GROUP - 8, 170):
        hay[p:p + 5] = b"abcde"
    for ov in (False, True):
        assert np.array_equal(cols(a.find(bytes(hay), overlapping=ov)), o.find_raw(bytes(hay), overlapping=ov))
    st = a.path_stats()
    assert st["hot_calls"] == 2 and st["dense_tiles"] == st["dense_radix"] == 0, st
    a.close()


def test_dense_everywhere_grows_the_overflow_list_and_stays_off_the_dense_path():
    # a pattern every 32 bytes EVERYWHERE (bench.py --dist D's shape): 128 hits per tile, twice the slots -- the overflow
    # lists are grown once for the call, every group is hot, and the hot pipeline is the whole post stage
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = gen.gen_uniform(8 << 20, gen.AZ, 12).copy()
    plant(hay, pats, 0, len(hay), 32, 77)
    hay = hay.tobytes()
    for mk in (0, 2):
        a = capi.Automaton(pats, mk, capi.IMPL_DFA)
        o = Oracle(pats, mk, KIND_DFA)
        want = o.find_raw(hay)
        a.path_stats(reset=True)
        assert np.array_equal(cols(a.find(hay)), want), mk
        st = a.path_stats(reset=True)
        assert st["hot_calls"] == 1 and st["overflow_regrown"] == 1 and st["dense_tiles"] == st["dense_radix"] == 0, st
        # (an input that is dense everywhere: the handle's next calls go to the dense path proper, no sparse attempt in front)
        assert np.array_equal(cols(a.find(hay)), want), mk
        st = a.path_stats()
        assert st["dense_tiles"] == 1 and st["hot_calls"] == 0 and st["overflow_regrown"] == 0, st
        # ... until an input that is not dense comes along: ONE call of it on the dense path, then the sparse kernels again
        # (until round 5 eight calls, each 2-3x the sparse path's time)
        plain = gen.gen_textlike(8 << 20, 12, pats).tobytes()
        want_plain = o.find_raw(plain)
        for expect in ("dense_tiles", "sparse", "sparse"):
            a.path_stats(reset=True)
            assert np.array_equal(cols(a.find(plain)), want_plain), (mk, expect)
            st = a.path_stats()
            assert st[expect] == 1 and sum(st[k] for k in ("sparse", "hot_calls", "dense_tiles", "dense_radix")) == 1, (expect, st)
        a.close()


def test_hot_groups_in_a_batch_and_with_code_points():
    # batch (segmented stream, local offsets + per-haystack counts taken by the write kernels) and the str API's
    # code-point conversion (the hot groups' words carry no chunk count: counted in place)
    pats = gen.gen_patterns(5000, 5, 12, gen.AZ, 3)
    o = Oracle(pats, 0, KIND_DFA)
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    hays = []
    for i in range(40):
        h = gen.gen_textlike(100_000 + 37 * i, 100 + i, pats).copy()
        if i % 7 == 3:
            plant(h, pats, 20_000, 60_000, 32, i)
        hays.append(h.tobytes())
    a.path_stats(reset=True)
    m, counts = a.find_batch(hays)
    st = a.path_stats()
    assert st["hot_calls"] == 1 and st["dense_tiles"] == st["dense_radix"] == 0, st
    at = 0
    for i, h in enumerate(hays):
        want = o.find_raw(h)
        assert counts[i] == len(want), i
        assert np.array_equal(cols(m[at:at + len(want)]), want), i
        at += len(want)
    a.close()
    spats = list(dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5)))
    bpats = [p.encode() for p in spats]
    host = gen.gen_unicode_textlike_bytes(3 << 20, 56, spats).copy()
    lo = 1 << 20
    while (host[lo] & 0xC0) == 0x80:
        lo += 1
    rng = gen.SplitMix64(9)
    k = lo
    while k < lo + (96 << 10):  # planted every ~20 bytes, at character boundaries (a character cut in half is blanked)
        p = np.frombuffer(bpats[rng.next() % len(bpats)], dtype=np.uint8)
        host[k:k + len(p)] = p
        k += len(p)
        while (host[k] & 0xC0) == 0x80:
            host[k] = 0x20
            k += 1
        k += 11
        while (host[k] & 0xC0) == 0x80:
            k += 1
    hay = host.tobytes()
    hay.decode("utf-8")
    b2c = byte_to_code_point(hay)
    for mk in (0, 2):
        a = capi.Automaton(bpats, mk)
        want = Oracle(bpats, mk, KIND_DFA).find_raw(hay)
        a.path_stats(reset=True)
        got = cols(a.find(hay, codepoints=True))
        st = a.path_stats()
        assert st["hot_calls"] == 1 and st["dense_tiles"] == st["dense_radix"] == 0, st
        assert np.array_equal(got[:, 0], want[:, 0])
        assert np.array_equal(got[:, 1], b2c[want[:, 1]]) and np.array_equal(got[:, 2], b2c[want[:, 2]])
        a.close()


def test_hot_groups_of_an_anchored_set():
    # patterns with common beginnings are filed under a rarer offset (anchors): a hit lies BEHIND the start of its
    # occurrence, the look-ahead tile of a hot group is filed too
    import random
    r = random.Random(5)
    pats = [b"http://www." + bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(4, 12))) for _ in range(3000)]
    pats = list(dict.fromkeys(pats))
    n = 4 * GROUP
    hay = np.frombuffer(bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz ./:") for _ in range(n)), dtype=np.uint8).copy()
    rng = gen.SplitMix64(3)
    for k in range(0, n - 64, 5000):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p
    k = GROUP + 60 * 4096  # a dense stretch over the boundary of groups 1 and 2
    while k < 2 * GROUP + 6 * 4096:
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p
        k += len(p) + 3
    check_all_kinds(pats, hay.tobytes(), True, impl=capi.IMPL_AUTO)
