"""GPU tests added in round 4 (each one answers an item of VERDICT.md / ADVICE.md, round 3):

  * pattern sets with patterns of 1 and 2 bytes stay on K1b (the side test): every kind, wide alphabets
    (> 32 byte classes: the sets that used to fall to the chunked walk), saturated level-1 tables, str API,
    batches (a short pattern never crosses a haystack boundary), unaligned buffers, sets of short patterns only
  * a batch of empty haystacks after a batch with matches returns zero counts (ADVICE, high)
  * RCCL behind the C ABI at one rank
  * anchors: patterns with common beginnings (a multi-byte character, "http://") are filed under a later
    offset; a hit lies behind the start of its occurrence -- the head is verified, haystack and group boundaries,
    the dense path, the str API
  * the tile-ordered dense path (occurrence buckets by key tile, sort + match kind in LDS) against the oracle and
    against the radix-sort form it replaces; its give-up cases (a bucket overflows, a chain leaves its context)
"""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def check_all_kinds(pats, hay, what, kernels=(None,)):
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        for kernel in kernels:
            a = capi.Automaton(pats, mk, kernel=kernel)
            if kernel is None:
                assert capi.KERNEL_NAMES[a.info.kernel] == "prefilter", what
            for ov in ([False, True] if mk == 0 else [False]):
                got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
                assert np.array_equal(got, want), (what, mk, ov, kernel, len(got), len(want))
            a.close()


SHORT_SETS = {
    "rare": [b"qz", b"~"],
    "frequent": [b"ab", b"x"],
    "readme": [b"b", b"abcd"],
    "dups": [b"q", b"qz", b"q", b"qz", b"zq"],
    "nul": [b"\0", b"a\0", b"\0\0"],
}


@pytest.mark.parametrize("name", list(SHORT_SETS))
def test_short_patterns_ride_the_prefilter(name):
    """cfg2's set + short patterns: K1b is the default kernel (round 3: the failureless walk), every
    kind equals the oracle; the haystack ends in a short pattern (the last positions are legal starts)."""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1) + SHORT_SETS[name]
    hay = gen.gen_textlike(3 << 20, 11, pats[:10000]).tobytes() + b"qzq\0a\0x~q\0"
    check_all_kinds(pats, hay, name)


def test_short_patterns_with_a_wide_alphabet():
    """> 32 byte classes AND a short pattern: the case that used to end on the chunked walk (0.5 TB/s with a
    dense table, 18 GB/s without).  cfg5's alphabet (a-z + 2/3/4-byte characters), mixed case + digits, all
    byte values."""
    uni = [p.encode() for p in dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5))]
    hay = gen.gen_unicode_textlike(1 << 20, 56, [p.decode() for p in uni]).encode()
    check_all_kinds(uni + ["é".encode(), b"Q", b"zq"], hay, "utf8")
    alnum = bytes(range(48, 58)) + bytes(range(65, 91)) + bytes(range(97, 123))
    pats = gen.gen_patterns(5000, 4, 10, alnum, 8) + [b"Zz", b"7", b"_"]
    hay = gen.gen_uniform(2 << 20, alnum + b" _", 9).tobytes()
    a = capi.Automaton(pats, 0)
    assert a.info.n_classes > 32 and capi.KERNEL_NAMES[a.info.kernel] == "prefilter"
    a.close()
    check_all_kinds(pats, hay, "alnum")
    pats = gen.gen_patterns(2000, 3, 9, gen.ALL_BYTES, 4) + [b"\xff", b"\x80\x81", b"\0"]
    hay = gen.gen_uniform(1 << 20, gen.ALL_BYTES, 3).tobytes()
    check_all_kinds(pats, hay, "bytes")


def test_short_patterns_only_and_tiny_sets():
    """No long pattern at all (empty prefilter tables), and sets whose long patterns are 3 and 4 bytes."""
    hay = gen.gen_textlike(1 << 20, 3).tobytes()
    for pats in ([b"q"], [b"qz"], [b"qz", b"zq", b"j", b"j"], [b"qz", b"qzx"], [b"~", b"qzxv", b"abc"]):
        check_all_kinds(pats, hay, pats, kernels=(None, capi.KERNEL_DFA_WALK))


def test_short_patterns_with_a_saturated_filter():
    """10^5 patterns (the BIG variant of K1b: both tests for every position, the bitmap stage) + short ones."""
    pats = gen.gen_patterns(100000, 5, 12, gen.AZ, 3) + [b"qz", b"~", b"Q"]
    hay = gen.gen_uniform(2 << 20, gen.AZ, 12).tobytes() + b"~Qqz"
    for mk, ov in ((0, True), (0, False), (2, False)):
        a = capi.Automaton(pats, mk)
        assert capi.KERNEL_NAMES[a.info.kernel] == "prefilter"
        got, want = cols(a.find(hay, overlapping=ov)), Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping=ov)
        assert np.array_equal(got, want), (mk, ov, len(got), len(want))
        a.close()


def test_short_patterns_in_batches_and_unaligned_buffers():
    """A 2-byte pattern never matches across two haystacks of a batch; unaligned device buffers shift the
    parity of the side test's pairs."""
    pats = gen.gen_patterns(2000, 5, 12, gen.AZ, 1) + [b"qz", b"z"]
    a = capi.Automaton(pats, 2)
    o = Oracle(pats, 2, KIND_DFA)
    rng = np.random.default_rng(5)
    hs = []
    for i in range(300):
        n = int(rng.integers(0, 9000))
        h = bytearray(gen.gen_textlike(n, 100 + i, pats[:2000]).tobytes())
        if n:
            h[-1] = ord("q")  # ... q | z ...: the pair straddles the boundary
        if n > 1 and i % 2:
            h[0] = ord("z")
        hs.append(bytes(h))
    m, counts = a.find_batch(hs)
    pos = 0
    for i, h in enumerate(hs):
        want = o.find_raw(h)
        assert np.array_equal(cols(m[pos:pos + int(counts[i])]), want), i
        pos += int(counts[i])
    assert pos == len(m)
    big = gen.gen_textlike((1 << 20) + 64, 7, pats[:2000])
    buf = capi.DeviceBuffer(len(big) + 64).upload(big)
    for lead in (0, 1, 2, 7, 15):
        n = (1 << 20) - 3
        r = a.find_device(buf.ptr + lead, n)
        got = cols(r.matches())
        r.free()
        assert np.array_equal(got, o.find_raw(big[lead:lead + n].tobytes())), lead
    buf.free()
    a.close()


def test_short_patterns_str_api():
    """The str API (code points, the CP variant of the scan) with a 1-character pattern of 1 and of 2 bytes."""
    import ahocorasick_rs_amd as ac
    spats = list(dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5))) + ["é", "Q", "zq"]
    hay = gen.gen_unicode_textlike(300000, 56, spats[:3000]) + "Qzqé"
    bpats = [p.encode() for p in spats]
    bhay = hay.encode()
    cp = np.cumsum(np.frombuffer(bhay, dtype=np.uint8) & 0xC0 != 0x80) - 1  # byte offset -> code point of its character
    cp = np.concatenate([cp, [cp[-1] + 1]])
    for mk_name, mk in (("Standard", 0), ("LeftmostLongest", 2)):
        a = ac.AhoCorasick(spats, matchkind=getattr(ac.MatchKind, mk_name))
        got = a.find_matches_as_indexes(hay)
        want = Oracle(bpats, mk, KIND_DFA).find_raw(bhay)
        assert got == [(int(p), int(cp[s]), int(cp[e])) for p, s, e in want], mk_name


def test_batch_of_empty_haystacks_after_a_batch_with_matches():
    """the per-haystack counts of a call that never reaches the pipeline (only empty
    haystacks) came out of the buffer cache holding the previous call's counts."""
    import ahocorasick_rs_amd as ac
    pats = [b"abc", b"bcd", b"zz"]
    a = capi.Automaton(pats, 0)
    m, counts = a.find_batch([b"abcd", b"zzz", b"xabc"])
    assert list(counts) == [1, 1, 1] and len(m) == 3
    for _ in range(3):
        m, counts = a.find_batch([b"", b"", b""])
        assert len(m) == 0 and list(counts) == [0, 0, 0]
    a.close()
    b = ac.BytesAhoCorasick(pats)
    assert b.find_matches_as_indexes_batch([b"abcd", b"zzz", b"xabc"]) == [[(0, 0, 3)], [(2, 0, 2)], [(0, 1, 4)]]
    for _ in range(3):
        assert b.find_matches_as_indexes_batch([b"", b""]) == [[], []]
    e = capi.Automaton([], 0)
    m, counts = e.find_batch([b"abc", b"", b"zz"])
    assert len(m) == 0 and list(counts) == [0, 0, 0]
    e.close()


def test_count_exchange_through_the_c_abi_one_rank():
    """RCCL behind the C ABI, executed on the MI355X at one rank (more needs a node): both ways of making the
    communicator -- ncclCommInitAll for one process, ncclGetUniqueId + ncclCommInitRank for one process per
    device -- all-gather a count and turn the counts into output offsets; then a sharded batch call whose
    count exchange goes through it.  In a process of its own, WITHOUT torch: that is the host this form exists
    for (a PyO3 shim) -- and a process into which a PyTorch wheel has been imported holds a second HIP / HSA
    runtime (the wheel bundles its own, under another SONAME), next to which /opt/rocm's RCCL does not initialise."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path[:0] = [%r, %r]
import gen
from ahocorasick_rs_amd import capi, distributed as D
assert "torch" not in sys.modules
c = capi.Comm.init_all([0])
assert (c.world, c.local_ranks) == (1, 1)
for n in (0, 42, 1 << 40):
    assert c.allgather_counts([n]) == [n]
c.close()
uid = capi.comm_unique_id()
assert len(uid) == capi.COMM_ID_BYTES
c = capi.Comm.init_rank(uid, 1, 0, 0)
pats = gen.gen_patterns(2000, 5, 12, gen.AZ, 1)
a = capi.Automaton(pats, 0)
hs = [gen.gen_textlike(5000, 40 + i, pats).tobytes() for i in range(64)]
lo, hi = capi.shard_range(len(hs), 0, c.world)
m, counts = a.find_batch(hs[lo:hi])
rank_counts, off, total = D.gather_match_counts_capi(c, len(m), 0)
assert rank_counts == [len(m)] and off == 0 and total == len(m) == int(counts.sum()) > 0
c.close()
a.close()
try:
    capi.Comm.init_all([0, 0])
    raise SystemExit("a device listed twice was accepted")
except ValueError:
    pass
assert "torch" not in sys.modules
print("RCCL_C_ABI_OK", total)
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_C_ABI_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])



def url_like_patterns(n=3000, seed=3):
    rng = np.random.default_rng(seed)
    hosts = [bytes(rng.integers(97, 123, int(rng.integers(3, 9))).astype(np.uint8)) for _ in range(n)]
    return [b"http://" + h + [b".com", b".org/x", b".net/index", b""][i % 4] for i, h in enumerate(hosts)] + [b"http://", b"https:"]


def test_anchored_patterns_cfg5_shape_and_urls():
    """The library files cfg5's emoji-led patterns and a URL list behind their common beginnings (max_shift > 0
    in the host tables); every kind equals the oracle, through both scan kernels (the walk does not use anchors)."""
    import ahocorasick_rs_amd as ac
    spats = list(dict.fromkeys(gen.gen_patterns(4000, 5, 12, gen.AZ_UNI, 5)))
    bpats = [p.encode() for p in spats]
    h = capi.HostAutomaton(bpats)
    assert int(h.t.max_shift) >= 3
    h.close()
    hay = gen.gen_unicode_textlike(1500000, 56, spats)
    bhay = hay.encode()
    check_all_kinds(bpats, bhay, "cfg5 shape", kernels=(None, capi.KERNEL_DFA_WALK))
    cp = np.cumsum(np.frombuffer(bhay, dtype=np.uint8) & 0xC0 != 0x80) - 1
    cp = np.concatenate([cp, [cp[-1] + 1]])
    a = ac.AhoCorasick(spats, matchkind=ac.MatchKind.LeftmostLongest)
    want = Oracle(bpats, 2, KIND_DFA).find_raw(bhay)
    assert a.find_matches_as_indexes(hay) == [(int(p), int(cp[s]), int(cp[e])) for p, s, e in want]
    urls = url_like_patterns()
    h = capi.HostAutomaton(urls)
    assert int(h.t.max_shift) >= 5
    h.close()
    text = bytearray(gen.gen_textlike(3 << 20, 9).tobytes())
    rng = np.random.default_rng(4)
    for o in rng.integers(0, len(text) - 40, 20000):
        u = urls[int(rng.integers(0, len(urls)))]
        u = u[:len(u) - int(rng.integers(0, 3))]  # (truncated copies: the anchor matches, the rest does not)
        text[o:o + len(u)] = u
    text[:len(urls[0])] = urls[0]           # an occurrence at the very start: its anchor lies 7 bytes in
    text[-len(urls[1]):] = urls[1]          # and one that ends the haystack
    check_all_kinds(urls, bytes(text), "urls")


@pytest.mark.parametrize("dense", [False, True])
def test_anchored_patterns_at_boundaries(monkeypatch, dense):
    """A hit lies up to 12 bytes behind the start of its occurrence: occurrences that start in the last bytes
    of a tile / of a 256 KiB group (their hits are in the next group's first tile), at the start of a haystack of
    a batch (the head must not reach into the previous haystack), unaligned buffers; sparse and dense path."""
    if dense:
        monkeypatch.setenv("ACX_NO_BUCKET", "1")
    urls = url_like_patterns(800, 5)
    rng = np.random.default_rng(8)
    hay = bytearray(gen.gen_textlike((1 << 20) + 4096, 21).tobytes())
    for edge in list(range(4096, len(hay) - 64, 4096 * 7)) + [262144, 524288, 786432, 1048576]:
        for back in (1, 3, 6, 7, 8, 11, 13):
            u = urls[int(rng.integers(0, len(urls)))]
            o = edge - back
            hay[o:o + len(u)] = u
    hay = bytes(hay)
    for mk in (0, 1, 2):
        a = capi.Automaton(urls, mk)
        o = Oracle(urls, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            assert np.array_equal(cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)), (mk, ov)
        big = np.frombuffer(hay, dtype=np.uint8)
        buf = capi.DeviceBuffer(len(big) + 64).upload(big)
        for lead, n in ((1, 300000), (9, 262144 + 4096), (15, 4096 * 3 - 15)):
            r = a.find_device(buf.ptr + lead, n)
            got = cols(r.matches())
            r.free()
            assert np.array_equal(got, o.find_raw(big[lead:lead + n].tobytes())), (mk, lead, n)
        buf.free()
        # batch: haystacks cut INSIDE planted URLs (the anchor "//xyz" opens a haystack whose head is elsewhere)
        cuts = sorted(set([0, len(hay)] + [int(x) for x in rng.integers(0, len(hay), 150)] + [4096 * 7 + 4096 - 3 + 7, 262144 - 7 + 5]))
        hs = [hay[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
        m, counts = a.find_batch(hs)
        pos = 0
        for i, hh in enumerate(hs):
            assert np.array_equal(cols(m[pos:pos + int(counts[i])]), o.find_raw(hh)), (mk, i)
            pos += int(counts[i])
        a.close()


def test_prose_and_word_text_64mib_against_the_oracle():
    """The reference benchmark's long shape at device scale (bench.py's config.secondary.prose): 4 244 names-like
    patterns over prose lines, and cfg2's set over word-shaped text -- a 64 MiB slice of each against the oracle,
    element-wise (the bench line tiles a 16 MiB period: four periods here, the seams included)."""
    names = [p.encode() for p in gen.names_like(4244, 6)]
    per = np.frombuffer(gen.names_haystack([p.decode() for p in names], 16 << 20, every=3), dtype=np.uint8)
    hay = np.tile(per, 4)
    for mk in (0, 2):
        a = capi.Automaton(names, mk)
        got = cols(a.find(hay))
        want = Oracle(names, mk, KIND_DFA).find_raw(hay.tobytes())
        assert len(want) > 30000 and np.array_equal(got, want), mk
        a.close()
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = np.tile(gen.gen_words(16 << 20, 11, pats), 4)
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    got = cols(a.find(hay))
    want = Oracle(pats, 0, KIND_DFA).find_raw(hay.tobytes())
    assert len(want) > 60000 and np.array_equal(got, want)
    a.close()


def dense_haystack(pats, n, every=32, seed=12):
    """uniform a-z with a pattern planted every `every` bytes (bench.py --dist D's shape)"""
    hay = gen.gen_uniform(n, gen.AZ, seed).copy()
    rng = gen.SplitMix64(77)
    for k in range(0, n - 32, every):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p
    return hay


@pytest.mark.parametrize("form", ["tiles", "radix"])
def test_dense_output_tile_ordered_and_radix_forms(monkeypatch, form):
    """Dense outputs (an occurrence every 32 bytes; cfg2's set + b"ab" + b"x": one every ~22 bytes) through the
    tile-ordered dense path and through the radix-sort form (ACX_NO_DENSE_TILES=1): every kind, overlapping,
    batches, unaligned buffers -- element-wise equal to the oracle."""
    if form == "radix":
        monkeypatch.setenv("ACX_NO_DENSE_TILES", "1")
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    hay = dense_haystack(pats, (6 << 20) + 77).tobytes()
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk)
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
            assert len(want) > 150000 and np.array_equal(got, want), (form, mk, ov, len(got), len(want))
        if mk == 2:
            big = np.frombuffer(hay, dtype=np.uint8)
            buf = capi.DeviceBuffer(len(big) + 64).upload(big)
            for lead, n in ((3, 1 << 20), (15, 4096 * 5 + 1)):
                r = a.find_device(buf.ptr + lead, n)
                got = cols(r.matches())
                r.free()
                assert np.array_equal(got, o.find_raw(big[lead:lead + n].tobytes())), (form, lead)
            buf.free()
            hs = [hay[i * 50001:(i + 1) * 50001] for i in range(40)] + [b"", hay[:9]]
            m, counts = a.find_batch(hs)
            pos = 0
            for i, hh in enumerate(hs):
                assert np.array_equal(cols(m[pos:pos + int(counts[i])]), o.find_raw(hh)), (form, i)
                pos += int(counts[i])
        a.close()
    mixedx = pats + [b"ab", b"x"]
    text = gen.gen_textlike(4 << 20, 11, pats).tobytes()
    for mk in (0, 2):
        a = capi.Automaton(mixedx, mk)
        got, want = cols(a.find(text)), Oracle(mixedx, mk, KIND_DFA).find_raw(text)
        assert len(want) > 100000 and np.array_equal(got, want), (form, mk)
        a.close()


def test_dense_tiles_str_api_and_give_up_cases():
    """Code points on the dense path (the str API over a dense haystack), and the inputs the tile-ordered form
    hands to the radix-sort form: more than one occurrence per 8 bytes (a bucket overflows), a chain of
    overlapping occurrences longer than the context tiles (periodic patterns on periodic text)."""
    import ahocorasick_rs_amd as ac
    spats = list(dict.fromkeys(gen.gen_patterns(3000, 5, 12, gen.AZ_UNI, 5)))
    hay = "".join(spats[(i * 7) % len(spats)] + "xé"[i % 2] for i in range(60000))
    bpats, bhay = [p.encode() for p in spats], hay.encode()
    cp = np.cumsum(np.frombuffer(bhay, dtype=np.uint8) & 0xC0 != 0x80) - 1
    cp = np.concatenate([cp, [cp[-1] + 1]])
    a = ac.AhoCorasick(spats, matchkind=ac.MatchKind.LeftmostLongest)
    want = Oracle(bpats, 2, KIND_DFA).find_raw(bhay)
    assert len(want) >= 60000
    assert a.find_matches_as_indexes(hay) == [(int(p), int(cp[s]), int(cp[e])) for p, s, e in want]
    # a bucket overflows: every second byte starts a match
    pats = [b"ab", b"abab", b"ba"]
    hay = b"ab" * (1 << 20)
    for mk in (0, 1, 2):
        a2 = capi.Automaton(pats, mk)
        for ov in ([False, True] if mk == 0 else [False]):
            assert np.array_equal(cols(a2.find(hay, overlapping=ov)), Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping=ov)), (mk, ov)
        a2.close()
    # a chain longer than the context: 3000-byte periodic patterns over the same period
    per = bytes(gen.gen_uniform(7, gen.AZ, 5))
    pats = [(per * 500)[i:i + 3000] for i in range(7)] + [per * 3]
    hay = per * 60000
    for mk in (1, 2):
        a3 = capi.Automaton(pats, mk)
        assert np.array_equal(cols(a3.find(hay)), Oracle(pats, mk, KIND_DFA).find_raw(hay)), mk
        a3.close()


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_single_haystack_cut_into_rank_ranges_device_resident(mk):
    """The same cut as tests/test_gpu_batch.py::test_single_haystack_cut_into_rank_ranges with the haystack
    resident in HBM (distributed.DeviceHaystack): the ranges are scanned where they lie -- unaligned range
    starts included -- and the resume after a changed carry reuses the first pass."""
    from ahocorasick_rs_amd import distributed as D
    pats = gen.gen_patterns(2000, 3, 9, gen.AZ, 41) + [b"abab", b"bab", b"ababab"]
    hay = bytearray(gen.gen_textlike((1 << 20) + 77, 42, pats).tobytes())
    for cut in range(1 << 17, 1 << 20, 1 << 17):
        hay[cut - 5:cut + 5] = b"ababababab"
    hay = bytes(hay)
    a = capi.Automaton(pats, mk)
    buf = capi.DeviceBuffer(len(hay)).upload(np.frombuffer(hay, dtype=np.uint8))
    try:
        for ov in ([False, True] if mk == 0 else [False]):
            whole = a.find(hay, overlapping=ov)
            for world in (2, 7):
                parts = D.simulate_single_sharded(a, D.DeviceHaystack(buf.ptr, len(hay)), world, overlapping=ov)
                got = np.concatenate(parts)
                assert got.tolist() == whole.tolist(), (mk, ov, world)
    finally:
        buf.free()
        a.close()


def test_byte_offsets_and_code_points_agree_on_ascii_and_contexts_survive_reuse():
    """The bytes API and the str API take different instantiations of the post kernels (offsets straight out /
    code points from the lead-byte counts); on ASCII text they must report the same numbers.  One handle, inputs of
    changing sizes, a call that leaves the sparse path in between: nothing of an older call may leak into a newer
    one (supergroup words, abort flags, hit counts)."""
    pats = gen.gen_patterns(3000, 3, 12, gen.AZ, 77) + [b"abab", b"bab", b"ababab"]
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk)
        o = Oracle(pats, mk, KIND_DFA)
        sizes = [(8 << 20) + 123, 1 << 20, (24 << 20) + 5, 300_000, 8 << 20]
        for k, n in enumerate(sizes):
            hay = gen.gen_textlike(n, 1000 + k, pats).tobytes()
            for ov in ([False, True] if mk == 0 else [False]):
                plain = a.find(hay, overlapping=ov)
                cps = a.find(hay, overlapping=ov, codepoints=True)  # ASCII: a code point is a byte
                assert plain.tolist() == cps.tolist(), (mk, n, ov)
                if n <= (1 << 20):
                    raw = o.find_raw(hay, ov)
                    assert np.stack([plain["pattern"], plain["start"], plain["end"]], 1).tolist() == raw.tolist()
            if k == 1:  # a call that leaves the sparse path (every byte a match) in between
                dense = a.find(b"ab" * (1 << 20), overlapping=False)
                assert len(dense) > 0
        a.close()


def test_groups_in_every_state():
    """Empty groups, full groups next to empty ones, matches across group borders (256 KiB) and in the very last
    bytes: the output offsets of the groups have to add up to the oracle's order."""
    pats = [b"needle", b"needles", b"edle", b"zzzz"]
    G = 64 * 4096
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk)
        o = Oracle(pats, mk, KIND_DFA)
        hay = bytearray(b"." * (9 * G + 777))
        for pos in (0, 5, G - 3, G, 2 * G - 6, 2 * G + 1, 5 * G - 1, 7 * G + 4000, len(hay) - 7, len(hay) - 6):
            hay[pos:pos + 7] = b"needles"[:min(7, len(hay) - pos)]
        hay[3 * G:3 * G + 4000] = b"needle" * 666 + b"need"  # one busy group between empty ones
        hay = bytes(hay)
        for ov in ([False, True] if mk == 0 else [False]):
            got = a.find(hay, overlapping=ov)
            raw = o.find_raw(hay, ov)
            assert np.stack([got["pattern"], got["start"], got["end"]], 1).tolist() == raw.tolist(), (mk, ov)
        a.close()


def test_anchored_set_with_a_pattern_longer_than_ten_kib():
    """Four context tiles in front of a group (patterns longer than 10 241 bytes) AND a look-ahead tile behind it
    (anchors): 69 tiles are read, but only 68 buckets exist -- k_tile_main once walked one row beyond its stage
    here (found reading the code in round 4; the ordering and sync-point phases now stop at the buckets)."""
    rng = np.random.default_rng(12)
    urls = url_like_patterns(600, 5)
    long_pat = bytes(rng.integers(97, 123, 11000, dtype=np.uint8))
    long2 = urls[3] + bytes(rng.integers(97, 123, 10500, dtype=np.uint8))  # a long one behind a crowded beginning
    pats = urls + [long_pat, long2]
    h = capi.HostAutomaton(pats)
    assert int(h.t.max_shift) > 0
    h.close()
    text = bytearray(gen.gen_textlike(6 << 20, 19).tobytes())
    for o in rng.integers(0, len(text) - 12000, 3000):
        u = pats[int(rng.integers(0, len(urls)))]
        text[o:o + len(u)] = u
    G = 64 * 4096
    for o in (G - 5000, 3 * G - 11000 - 3, 5 * G + 17, 7 * G - 10, len(text) - 11000):
        text[o:o + 11000] = long_pat
    text[9 * G - 40:9 * G - 40 + len(long2)] = long2
    check_all_kinds(pats, bytes(text), "anchors + 11 000-byte pattern")


def _k0_sets():
    """Automata that take each of K0's three ways of finding the occurrences (kernels.hip, k0_small<MODE>)."""
    few = [b"abc", b"hello", b"b", b"aardvark", b"fish", b"whatwhat", b"sixteen-bytes-xy", b"ninebytes", b"host7", b"host76", b"b"]
    return {
        "direct comparison": few,                                   # <= 64 patterns of <= 16 bytes
        "table in LDS": few + [b"seventeen-bytes-xy"],              # a longer pattern: the walk, small tables
        "tables in global memory": few + gen.gen_patterns(3000, 4, 12, gen.AZ, 3),
    }


@pytest.mark.parametrize("which", ["direct comparison", "table in LDS", "tables in global memory"])
def test_k0_every_mode_every_kind_result_line_and_overflow(which):
    """One call on a small host haystack = K0.  0, 1, 5 (the result line is full), 6 and 7 (the first packed
    records beyond the line), hundreds of matches (SMALL_MAX_OCC is 1024: beyond it the general pipeline), matches
    at the very start and end, windows at every alignment, a haystack of one byte and the empty one -- bytes
    and code points, every kind, against the oracle."""
    import ahocorasick_rs_amd as ac
    pats = _k0_sets()[which]
    hays = [b"", b"b", b"x", b"hello", b"xhellox", b"fish host76 abc", b"abc" * 2, b"b" * 5, b"b" * 6, b"b" * 7,
            b"hello fish abc b host7 host76 whatwhat sixteen-bytes-xy ninebytes aardvark",
            b"b" * 300, b"ab" * 600, b"sixteen-bytes-xy" * 20 + b"sixteen-bytes-x", b"q" * 407 + b"ninebytes"]
    hays += [b"z" * k + b"seventeen-bytes-xy" + b"z" * (9 - k) + b"aardvark" for k in range(9)]
    kinds = [(ac.MatchKind.Standard, 0), (ac.MatchKind.LeftmostFirst, 1), (ac.MatchKind.LeftmostLongest, 2)]
    for mkind, mk in kinds:
        o = Oracle(pats, mk, KIND_DFA)
        a = ac.BytesAhoCorasick(pats, matchkind=mkind)
        for hay in hays:
            for ov in ([False, True] if mk == 0 else [False]):
                want = [tuple(int(x) for x in r) for r in o.find_raw(hay, ov)]
                assert a.find_matches_as_indexes(hay, overlapping=ov) == want, (which, mk, ov, hay[:40])
    # code points: the same through the str API with two- and four-byte characters in front of and between the matches
    spats = [p.decode() for p in pats] + ["é", "🤦b"]
    bpats = [p.encode() for p in spats]
    text = "é🤦b hello ☃ fish é abc 🤦 host76 ééé b" + "é" * 40 + "whatwhat"
    bts = text.encode()
    cp = np.cumsum(np.frombuffer(bts, dtype=np.uint8) & 0xC0 != 0x80) - 1
    cp = np.concatenate([cp, [cp[-1] + 1]])
    for mkind, mk in kinds:
        a = ac.AhoCorasick(spats, matchkind=mkind)
        want = [(int(p), int(cp[s]), int(cp[e])) for p, s, e in Oracle(bpats, mk, KIND_DFA).find_raw(bts)]
        assert a.find_matches_as_indexes(text) == want, (which, mk)


def test_k0_direct_comparison_work_limit():
    """haystack bytes x patterns <= 4096 takes the direct comparison, one byte more the walk: the same answers on
    both sides of the limit (64 patterns: haystacks of 64 and 65 bytes; 10 patterns: 409 and 410)."""
    import ahocorasick_rs_amd as ac
    rng = np.random.default_rng(5)
    for npat in (64, 10):
        pats = list(dict.fromkeys(bytes(rng.integers(97, 101, int(rng.integers(1, 17))).astype(np.uint8)) for _ in range(npat * 3)))[:npat]
        o = Oracle(pats, 2, KIND_DFA)
        a = ac.BytesAhoCorasick(pats, matchkind=ac.MatchKind.LeftmostLongest)
        for n in (4096 // npat - 1, 4096 // npat, 4096 // npat + 1, 4096 // npat + 17):
            hay = bytes(rng.integers(97, 101, n).astype(np.uint8))
            want = [tuple(int(x) for x in r) for r in o.find_raw(hay, False)]
            assert a.find_matches_as_indexes(hay) == want, (npat, n)


@pytest.mark.parametrize("kernel", [None, capi.KERNEL_DFA_WALK])
def test_copies_of_a_pattern_cost_a_non_overlapping_search_nothing(kernel):
    """Hundreds of copies of every string (tools/gpu_fuzz.py seed 40404: 22 264 patterns = 30 distinct strings) on
    text where every position matches: a non-overlapping search reports the lowest id of a string and must not pay
    for the others (acx_api.cpp: dev_nov, the view without the later copies); an overlapping one reports every copy.
    Short strings (own lists / K0), longer ones (K1b's candidate lists), 1- and 2-byte ones (the side test's lists)."""
    import random
    import time
    rng = random.Random(7)
    sets = {
        "1-4 bytes": [bytes(rng.choice(b"ab") for _ in range(rng.randint(1, 4))) for _ in range(6000)],
        "5-8 bytes": [bytes(rng.choice(b"ab") for _ in range(rng.randint(5, 8))) for _ in range(20000)],
        "3-6 bytes": [bytes(rng.choice(b"abc") for _ in range(rng.randint(3, 6))) for _ in range(30000)],
    }
    hay = bytes(rng.choice(b"ab") for _ in range(300_000)) + bytes(rng.choice(b"abc") for _ in range(300_000))
    for name, pats in sets.items():
        copies = len(pats) / len(set(pats))
        assert copies > 20, (name, copies)
        o = Oracle(pats, 0, KIND_DFA)
        a = capi.Automaton(pats, 0, kernel=kernel)  # (a forced scan kernel also keeps K0 out: every size through the pipeline)
        for h in (hay, hay[:3000], hay[299_000:301_500]):  # the pipeline; K0; K0 across the alphabet change
            want = o.find_raw(h, False)
            got = cols(a.find(h))
            t0 = time.time()
            got2 = cols(a.find(h))
            dt = time.time() - t0
            assert np.array_equal(got, want) and np.array_equal(got2, want), (name, len(h), len(got), len(want))
            assert dt < 1.0, (name, len(h), dt)  # (the forced DFA walk took 28 s for 20 000 bytes of this)
        small = hay[:1500]
        assert np.array_equal(cols(a.find(small, overlapping=True)), o.find_raw(small, True)), name  # every copy
        a.close()
