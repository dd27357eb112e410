"""GPU: the seams between the execution paths -- the sparse output path (per-tile hit slots +
k_tile_main): greedy chains that cross tile and group boundaries, chains longer than a group's
context, slot / bucket / group overflow into the dense (region + radix sort) path, the hold-off
after a dense call, long patterns (context tiles), two-level prefix keys; and K0, the
one-workgroup kernel that answers small haystacks."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
KERNELS = [capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER]
TILE = 64 * 4096  # bytes of index space per group of k_tile_main (GROUP_TILES * 4 KiB)


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def check(a, o, hay, mk):
    for ov in ([False, True] if mk == 0 else [False]):
        assert np.array_equal(cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)), (mk, ov)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_chains_across_bucket_and_tile_boundaries(mk, kernel):
    # runs of mutually overlapping occurrences laid over every kind of boundary: the greedy
    # chain of a group then starts in its context tiles (k_tile_main re-derives it from the
    # nearest certified sync point)
    pats = [b"ababab", b"babab", b"abab", b"bab", b"abababababab"]
    hay = bytearray(b"x" * (3 * TILE + 5000))
    run = b"ab" * 14  # 28 bytes, < 32 occurrences per bucket
    for centre in (4096, 8192 + 1, TILE, 2 * TILE - 3, 2 * TILE + 4096, 3 * TILE + 2):
        for shift in (-27, -14, -1, 0):
            s = centre + shift
            hay[s:s + len(run)] = run
    # a chain that spans a whole bucket border region from far before: long pattern
    hay[TILE - 11:TILE + 1] = b"abababababab"
    hay = bytes(hay)
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    check(a, o, hay, mk)
    # an unaligned view of the same bytes (K1b's 16-byte lead, bucket of the key position)
    check(a, o, hay[7:], mk)
    a.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_bucket_and_tile_overflow_fall_back_to_dense_path(kernel):
    pats = [b"abcde", b"cde"]
    a = capi.Automaton(pats, 0, kernel=kernel)
    o = Oracle(pats, 0, KIND_DFA)
    # (1) one 4 KiB bucket with more than 32 occurrences in an otherwise empty stream
    hay = bytearray(b"." * (1 << 20))
    hay[50000:50000 + 5 * 40] = b"abcde" * 40
    check(a, o, bytes(hay), 0)
    # the calls right after run in region mode (hold-off), then the sparse path returns
    sparse = bytearray(b"." * (1 << 20))
    for p in range(1000, len(sparse) - 10, 3001):
        sparse[p:p + 5] = b"abcde"
    sparse = bytes(sparse)
    for _ in range(12):
        check(a, o, sparse, 0)
    # (2) no bucket above 32, but a tile above 1024: 24 occurrences in every bucket (48 with overlapping)
    hay = bytearray(b"." * (2 * TILE))
    for p in range(0, len(hay) - 8, 170):
        hay[p:p + 5] = b"abcde"
    check(a, o, bytes(hay), 0)
    check(a, o, sparse, 0)
    a.close()


def test_textlike_10k_patterns_matches_oracle_on_both_paths():
    # the headline automaton on 8 MiB of text-like data: sparse path, then forced region mode
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = gen.gen_textlike(8 << 20, 11, pats).tobytes()
    o = Oracle(pats, 0, KIND_DFA)
    want = o.find_raw(hay)
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    assert np.array_equal(cols(a.find(hay)), want)
    a.close()
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import gen; from ahocorasick_rs_amd import capi\n"
        "pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)\n"
        "hay = gen.gen_textlike(8 << 20, 11, pats).tobytes()\n"
        "a = capi.Automaton(pats, 0, capi.IMPL_DFA)\n"
        "m = a.find(hay)\n"
        "np.save(sys.argv[1], np.stack([m['pattern'], m['start'], m['end']], 1))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for env in ({"ACX_NO_BUCKET": "1"},):
        out = "/tmp/acx_sparse_path_%s.npy" % "_".join(env)
        subprocess.run([sys.executable, "-c", code, out], check=True, env={**os.environ, **env}, timeout=600)
        assert np.array_equal(np.load(out), want), env


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_chain_longer_than_the_context_takes_the_dense_path(mk, kernel):
    # a 256-byte pattern of period 128 on a periodic haystack: an occurrence every 128 bytes (32 per
    # tile: the slots hold them), each overlapping the next, no sync point anywhere -- no group can
    # certify where the greedy chain stands, the call is redone on the dense path
    unit = bytes((i * 7 + 3) % 251 for i in range(128))
    pats = [unit * 2, unit[5:60]]
    hay = b"q" * 1000 + unit * 4200 + b"q" * 777
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    check(a, o, hay, mk)
    check(a, o, hay[3:], mk)
    a.close()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("maxlen", [2049, 2050, 4096, 6200, 14000, 20000])
def test_long_patterns_use_more_context_tiles_or_the_dense_path(maxlen, kernel):
    # the context in front of a group must be longer than the longest pattern: 1, 2, .. tiles;
    # beyond MAX_LOOKBACK tiles the automaton stays on the dense path
    rng = np.random.default_rng(maxlen)
    long_p = rng.integers(97, 123, maxlen, dtype=np.uint8).tobytes()
    pats = [long_p, long_p[100:140], b"needle", long_p[-9:]]
    hay = bytearray(rng.integers(97, 123, 3 * TILE + 999, dtype=np.uint8).tobytes())
    for at in (5, TILE - maxlen // 2, 2 * TILE - 3, 2 * TILE + 4096 - maxlen + 1, 3 * TILE - maxlen):
        at = max(at, 0)
        hay[at:at + maxlen] = long_p
    hay[TILE + 50:TILE + 56] = b"needle"
    hay = bytes(hay)
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk, kernel=kernel)
        check(a, Oracle(pats, mk, KIND_DFA), hay, mk)
        a.close()


@pytest.mark.parametrize("kernel", KERNELS)
def test_slot_bucket_and_group_overflow(kernel):
    o_pats = [b"abcdefg", b"efg", b"g"]
    a = capi.Automaton(o_pats, 0, kernel=kernel)
    o = Oracle(o_pats, 0, KIND_DFA)
    # (1) more than 32 prefix hits in one tile; (2) 32 hits or fewer per tile, but more than 32
    # occurrences in one bucket of the END position (overlapping: three per hit); (3) no bucket
    # above 32, but more than 1024 reported matches in one group
    for step, span in ((7, 7 * 40), (300, 20 * 300), (160, TILE)):
        hay = bytearray(b"." * (2 * TILE))
        for p in range(TILE // 2, TILE // 2 + span, step):
            hay[p:p + 7] = b"abcdefg"
        check(a, o, bytes(hay), 0)
    a.close()


def test_two_level_prefix_keys_on_the_device():
    # patterns of 5 bytes next to longer ones sharing 5..7 bytes, zero bytes inside the keys, duplicates
    pats = [b"abcde", b"abcde\0", b"abcde\0\0\0", b"abcdefgh", b"abcdefgi", b"abcdefghij", b"abcdf\0x", b"abcdf\0y",
            b"abcde", b"zzzzzzzzzzzz", "\U0001F926a-tail".encode(), "\U0001F926b-tail".encode(), b"xyzzy"]
    rng = np.random.default_rng(5)
    pieces = [bytes(p) for p in pats] + [b"abcd", b"abcdf\0", "\U0001F926".encode(), b"\0\0", b"abcdefg", b"zzzzzzz"]
    hay = b"".join(pieces[i] + bytes(rng.integers(0, 3, rng.integers(0, 4), dtype=np.uint8)) for i in
                   rng.integers(0, len(pieces), 40000))
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk, kernel=capi.KERNEL_PREFILTER)
        check(a, Oracle(pats, mk, KIND_DFA), hay, mk)
        a.close()


# ---------------------------------------------------------------------------
# K0: small haystacks answered by one workgroup
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_small_call_kernel_equals_general_pipeline_and_oracle(mk):
    import random
    rng = random.Random(900 + mk)
    for trial in range(12):
        alpha = rng.choice([b"ab", b"abcd", bytes(range(97, 123)), bytes(range(256))])
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, rng.choice([3, 8, 40]))))
                for _ in range(rng.choice([1, 5, 60, 400]))]
        pats += [pats[0]]  # a duplicate pattern: ties resolve to the lowest index
        n = rng.choice([1, 2, 15, 16, 17, 255, 1000, 4097, 16383, 16384])
        hay = bytearray(rng.choice(alpha) for _ in range(n))
        for _ in range(rng.randint(0, 6)):  # plant whole patterns, also flush with both ends
            p = rng.choice(pats)
            if len(p) <= n:
                at = rng.choice([0, n - len(p), rng.randint(0, n - len(p))])
                hay[at:at + len(p)] = p
        hay = bytes(hay)
        auto = capi.Automaton(pats, mk)                                   # small haystack -> K0
        forced = capi.Automaton(pats, mk, kernel=capi.KERNEL_DFA_WALK)    # explicit kernel: never K0
        o = Oracle(pats, mk, KIND_DFA)
        n_occ = len(Oracle(pats, 0, KIND_DFA).find_raw(hay, overlapping=True))  # what K0 has to hold
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            auto.profile_read(reset=True)
            got = cols(auto.find(hay, overlapping=ov))
            k0_calls = auto.profile_read().small_calls
            assert np.array_equal(got, want), (mk, ov, trial)
            assert np.array_equal(cols(forced.find(hay, overlapping=ov)), want), (mk, ov, trial)
            assert forced.profile_read().small_calls == 0
            assert k0_calls == (1 if n_occ <= 1024 else 0)  # dense output falls through to the pipeline
        auto.close()
        forced.close()


def test_small_call_kernel_limits_and_device_entry():
    pats = [b"a", b"aa", b"abc"]
    a = capi.Automaton(pats, 0)
    o = Oracle(pats, 0, KIND_DFA)
    # 16 KiB is the last K0 size, 16 KiB + 1 the first size of the general pipeline
    for n, small in ((16384, 1), (16385, 0)):
        hay = ((b"x" * 40 + b"abc") * 400)[:n]
        a.profile_read(reset=True)
        assert np.array_equal(cols(a.find(hay)), o.find_raw(hay))
        assert a.profile_read().small_calls == small
    # more than 1024 occurrences: K0 gives up, the general pipeline answers
    hay = b"a" * 3000
    for ov in (False, True):
        a.profile_read(reset=True)
        assert np.array_equal(cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov))
        assert a.profile_read().small_calls == 0
    # device-resident haystack and result
    hay = b"..abc..aa..a" * 100
    buf = capi.DeviceBuffer(len(hay) + 3)
    buf.upload(np.frombuffer(b"###" + hay, dtype=np.uint8))
    a.profile_read(reset=True)
    r = a.find_device(buf.ptr + 3, len(hay))  # unaligned device pointer
    assert np.array_equal(cols(r.matches()), o.find_raw(hay))
    assert a.profile_read().small_calls == 1
    r.free()
    a.close()


def test_small_call_kernel_code_points():
    ac = pytest.importorskip("ahocorasick_rs_amd")
    pats = ["é", "☃", "🤦", "aé", "z☃z", "needle"]
    hay = ("aé☃🤦z☃z" * 50 + "needle" + "é" * 7 + "🤦needle☃") * 3
    assert len(hay.encode()) <= 16384
    kinds = {0: ac.MatchKind.Standard, 1: ac.MatchKind.LeftmostFirst, 2: ac.MatchKind.LeftmostLongest}
    for mk, kind in kinds.items():
        a = ac.AhoCorasick(pats, matchkind=kind)
        got = a.find_matches_as_indexes(hay)
        assert got == Oracle([p.encode() for p in pats], mk, KIND_DFA).find_str(hay)
        for pid, s, e in got:  # offsets are code-point indexes: slicing the str gives the pattern
            assert hay[s:e] == pats[pid]


def test_profile_hooks_every_call_and_sampled():
    # acx_profile_enable(a, 1): every call carries the event pair around its scan kernel;
    # N > 1: every N-th call of a context; the totals cover the measured calls only
    pats = gen.gen_patterns(2000, 5, 9, gen.AZ, 77)
    hay = gen.gen_textlike(1 << 20, 78, pats, 2048).tobytes()
    a = capi.Automaton(pats, 0, kernel=capi.KERNEL_PREFILTER)
    o = Oracle(pats, 0, KIND_DFA)
    want = o.find_raw(hay, overlapping=False)
    for every, calls, measured in ((1, 6, 6), (4, 8, 2), (3, 7, 3)):
        a.profile_enable(every)
        a.profile_read(reset=True)
        for _ in range(calls):
            assert np.array_equal(cols(a.find(hay)), want)
        p = a.profile_read(reset=True)
        # (the sampling phase continues across enable calls: between floor and ceil of calls / every)
        assert calls // every <= p.scan_launches <= -(-calls // every), (every, p.scan_launches)
        if every == 1:
            assert p.scan_launches == measured
        assert p.scan_bytes == p.scan_launches * len(hay) and p.scan_ms > 0
        assert p.raw_occurrences >= p.scan_launches * len(want)
    a.profile_enable(False)
    a.profile_read(reset=True)
    assert np.array_equal(cols(a.find(hay)), want)
    assert a.profile_read().scan_launches == 0
    a.close()


@pytest.mark.parametrize("kernel", KERNELS + [None])
def test_dense_output_concentrated_in_one_place(kernel):
    # millions of occurrences in the first fifth of the haystack, (almost) none elsewhere: the regions
    # of the dense path fill very unevenly -- its second pass gives every region exactly the room its
    # first pass counted (capacity = grid x the fullest region would be two orders of magnitude more)
    import random
    rng = random.Random(5)
    pats = list({bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 8))) for _ in range(3000)})
    dense = bytes(rng.choice(b"abc") for _ in range(400_000))
    hay = dense + b"z" * 1_600_000
    for mk in (1, 0):
        a = capi.Automaton(pats, mk, kernel=kernel)
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            want = o.find_raw(hay, overlapping=ov)
            assert len(want) > 100_000
            assert np.array_equal(cols(a.find(hay, overlapping=ov)), want), (mk, ov)
        # and the handle still answers an ordinary call afterwards
        small = b"zzabcabczz" * 100
        assert np.array_equal(cols(a.find(small)), o.find_raw(small, overlapping=False))
        a.close()
