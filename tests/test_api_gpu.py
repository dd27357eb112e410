"""GPU tests of the reference-shaped API (the C++ extension over the C ABI).

Mirrors the behaviours pinned by the reference's own suites
(/root/reference/tests/test_ac.py, tests/test_ac_bytes.py) -- rewritten here,
not copied -- and checks the committed golden fixtures (reference known-answer
vectors, genuine-crate LeftmostLongest, `re` LeftmostFirst) through the HIP path.
"""
import json
import os
import random

import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from spec import spec

pytestmark = pytest.mark.gpu
ac = pytest.importorskip("ahocorasick_rs")  # the drop-in name (alias of ahocorasick_rs_amd)
from ahocorasick_rs import (AhoCorasick, BytesAhoCorasick, Implementation,  # noqa: E402
                            MatchKind, MATCHKIND_LEFTMOST_FIRST,
                            MATCHKIND_LEFTMOST_LONGEST, MATCHKIND_STANDARD)

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
IMPLS = [None, Implementation.NoncontiguousNFA, Implementation.ContiguousNFA, Implementation.DFA]
KINDS = [MatchKind.Standard, MatchKind.LeftmostFirst, MatchKind.LeftmostLongest]
FAST = settings(max_examples=60, deadline=None,
                suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


def load(name):
    with open(os.path.join(G, name), encoding="utf-8") as f:
        return json.load(f)


# ------------------------------------------------------------------ str API
@pytest.mark.parametrize("store_patterns", [True, False, None])
@pytest.mark.parametrize("implementation", IMPLS)
def test_str_basic(store_patterns, implementation):
    hay, pats = "hello, world, hello again", ["hello", "world"]
    a = AhoCorasick(pats) if store_patterns is None else AhoCorasick(
        pats, store_patterns=store_patterns, implementation=implementation)
    idx = a.find_matches_as_indexes(hay)
    assert idx == [(0, 0, 5), (1, 7, 12), (0, 14, 19)]
    assert all(isinstance(x, int) for t in idx for x in t) and isinstance(idx[0], tuple)
    assert a.find_matches_as_strings(hay) == ["hello", "world", "hello"]


@pytest.mark.parametrize("store_patterns", [True, False, None])
def test_str_patterns_from_iterator(store_patterns):
    kw = {} if store_patterns is None else {"store_patterns": store_patterns}
    a = AhoCorasick(iter(["hello", "world"]), **kw)
    assert a.find_matches_as_strings("hello, world, hello again") == ["hello", "world", "hello"]
    a = AhoCorasick((p.lower() for p in ["HELLO", "World"]), **kw)
    assert a.find_matches_as_strings("hello world") == ["hello", "world"]


@pytest.mark.parametrize("store_patterns", [True, False, None])
@pytest.mark.parametrize("implementation", IMPLS)
def test_str_unicode_codepoint_indexes(store_patterns, implementation):
    hay, pats = "hello, world ☃fishá l🤦l", ["d ☃f", "há", "l🤦l"]
    a = AhoCorasick(pats) if store_patterns is None else AhoCorasick(
        pats, store_patterns=store_patterns, implementation=implementation)
    idx = a.find_matches_as_indexes(hay)
    assert idx == [(0, 11, 15), (1, 17, 19), (2, 20, 23)]
    assert [hay[s:e] for (_, s, e) in idx] == pats
    assert a.find_matches_as_strings(hay) == pats


def test_matchkinds_and_aliases():
    hay = "This is the winter of my discontent"
    pats = ["content", "disco", "disc", "discontent", "winter"]
    for mk in (None, MATCHKIND_STANDARD, MatchKind.Standard):
        a = AhoCorasick(pats) if mk is None else AhoCorasick(pats, matchkind=mk)
        assert a.find_matches_as_strings(hay) == ["winter", "disc"]
    for mk in (MATCHKIND_LEFTMOST_FIRST, MatchKind.LeftmostFirst):
        assert AhoCorasick(pats, matchkind=mk).find_matches_as_strings(hay) == ["winter", "disco"]
    for mk in (MATCHKIND_LEFTMOST_LONGEST, MatchKind.LeftmostLongest):
        assert AhoCorasick(pats, matchkind=mk).find_matches_as_strings(hay) == ["winter", "discontent"]
    hb = hay.encode()
    pb = [p.encode() for p in pats]

    def strs(a):
        return [hb[s:e] for (_, s, e) in a.find_matches_as_indexes(hb)]
    assert strs(BytesAhoCorasick(pb)) == [b"winter", b"disc"]
    assert strs(BytesAhoCorasick(pb, matchkind=MatchKind.LeftmostFirst)) == [b"winter", b"disco"]
    assert strs(BytesAhoCorasick(pb, MatchKind.LeftmostLongest)) == [b"winter", b"discontent"]


def test_overlapping_and_its_error():
    hay = "This is the winter of my discontent"
    pats = ["content", "disco", "disc", "discontent", "winter"]
    want = ["winter", "disc", "disco", "discontent", "content"]
    for a in (AhoCorasick(pats), AhoCorasick(pats, matchkind=MatchKind.Standard)):
        assert a.find_matches_as_strings(hay) == a.find_matches_as_strings(hay, overlapping=False)
        assert a.find_matches_as_indexes(hay) == a.find_matches_as_indexes(hay, overlapping=False)
        res = a.find_matches_as_strings(hay, overlapping=True)
        idx = a.find_matches_as_indexes(hay, overlapping=True)
        assert res == want and [pats[i] for (i, _, _) in idx] == want
        assert [hay[s:e] for (_, s, e) in idx] == want
    for mk in (MatchKind.LeftmostFirst, MatchKind.LeftmostLongest):
        a = AhoCorasick(pats, matchkind=mk)
        with pytest.raises(ValueError):
            a.find_matches_as_strings(hay, overlapping=True)
        with pytest.raises(ValueError) as e:
            a.find_matches_as_indexes(hay, overlapping=True)
        assert "does not support overlapping" in str(e.value)
        b = BytesAhoCorasick([p.encode() for p in pats], matchkind=mk)
        with pytest.raises(ValueError):
            b.find_matches_as_indexes(hay.encode(), overlapping=True)
    b = BytesAhoCorasick([p.encode() for p in pats])
    hb = hay.encode()
    assert [hb[s:e].decode() for (_, s, e) in b.find_matches_as_indexes(hb, overlapping=True)] == want


def test_haystack_type_errors():
    a = AhoCorasick(["x"])
    with pytest.raises(TypeError):
        a.find_matches_as_indexes(b"x")
    b = BytesAhoCorasick([b"x"])
    with pytest.raises(TypeError):
        b.find_matches_as_indexes("x")
    with pytest.raises(TypeError):
        b.find_matches_as_indexes(12)
    import numpy as np
    with pytest.raises(TypeError) as e:
        b.find_matches_as_indexes(np.zeros((2, 2), dtype=np.uint8))
    assert "one-dimensional" in str(e.value)
    with pytest.raises(TypeError) as e:
        b.find_matches_as_indexes(memoryview(b"abcdef")[::2])
    assert "contiguous" in str(e.value)


def test_argument_types_follow_pyo3():
    """`overlapping: bool` (src/lib.rs:229, 253, 422) takes a real bool; the buffer must hold
    unsigned bytes (PyBuffer::<u8>::get, src/lib.rs:286)."""
    import array
    a, b = AhoCorasick(["x"]), BytesAhoCorasick([b"x"])
    for bad in (1, 0, "yes", None):
        with pytest.raises(TypeError):
            a.find_matches_as_indexes("x", overlapping=bad)
        with pytest.raises(TypeError):
            a.find_matches_as_strings("x", overlapping=bad)
        with pytest.raises(TypeError):
            b.find_matches_as_indexes(b"x", overlapping=bad)
    assert a.find_matches_as_indexes("x", True) == [(0, 0, 1)] == b.find_matches_as_indexes(b"x", overlapping=True)
    with pytest.raises(BufferError):
        b.find_matches_as_indexes(array.array("b", [120]))
    assert b.find_matches_as_indexes(array.array("B", [120])) == [(0, 0, 1)]


def test_concurrent_searches_from_threads():
    """The reference releases the GIL around matching and lets threads search ONE automaton
    concurrently (src/lib.rs:238, 261, 433; module gil_used = false, :438).  N threads on one
    object and N threads on N objects return exactly what a single thread returns."""
    import threading
    import gen
    from oracle_lib import KIND_DFA, Oracle
    pats = gen.gen_patterns(3000, 4, 10, gen.AZ, 71)
    hays = [gen.gen_textlike(n, 72 + i, pats, plant_every=256).tobytes()
            for i, n in enumerate([300, 5000, 70_000, 1 << 20, 3 << 20, 17, 0, 40_000])]
    o = Oracle(pats, 0, KIND_DFA)
    want = [o.find(h) for h in hays]
    want_ov = [o.find(h, overlapping=True) for h in hays]
    shared = BytesAhoCorasick(pats)
    own = [BytesAhoCorasick(pats) for _ in range(4)]
    errors = []

    def worker(t, automaton):
        try:
            for rep in range(6):
                for k in range(len(hays)):
                    i = (k + t + rep) % len(hays)
                    for ov, expect in ((False, want[i]), (True, want_ov[i])):
                        got = automaton.find_matches_as_indexes(hays[i], overlapping=ov)
                        if got != expect:  # (what differs, for the record: a mismatch here has been seen once in ~30 runs)
                            d = next((k for k, (x, y) in enumerate(zip(got, expect)) if x != y), min(len(got), len(expect)))
                            errors.append(("overlapping" if ov else "non-overlapping", t, i, len(hays[i]), len(got), len(expect), d,
                                           got[max(0, d - 1):d + 2], expect[max(0, d - 1):d + 2]))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    for automata in ([shared] * 8, own):
        ts = [threading.Thread(target=worker, args=(t, a)) for t, a in enumerate(automata)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors[:5]


# ---------------------------------------------------------------- bytes API
@pytest.mark.parametrize("implementation", IMPLS)
@pytest.mark.parametrize("haystack_type", [bytes, bytearray, memoryview])
def test_bytes_buffer_types(implementation, haystack_type):
    hay = haystack_type(b"hello, world, hello again")
    for pats in ([b"hello", b"world"], [memoryview(b"hello"), bytearray(b"world")]):
        a = BytesAhoCorasick(pats, implementation=implementation)
        idx = a.find_matches_as_indexes(hay)
        assert idx == [(0, 0, 5), (1, 7, 12), (0, 14, 19)]
    a = BytesAhoCorasick(iter([b"hello", b"world"]))
    assert a.find_matches_as_indexes(b"hello world") == [(0, 0, 5), (1, 6, 11)]


# ------------------------------------------------------- property tests
@FAST
@given(st.lists(st.text(min_size=3), min_size=1, max_size=300),
       st.sampled_from([True, False, None]))
def test_str_construction_extensive(patterns, store_patterns):
    patterns = [f"{p}_{i}_" for (i, p) in enumerate(patterns)]
    a = AhoCorasick(patterns, store_patterns=store_patterns)
    got = a.find_matches_as_indexes_batch(patterns)
    for p, g in zip(patterns, got):
        assert [p[s:e] for (_, s, e) in g] == [p]
    for p in patterns[:3]:
        assert a.find_matches_as_strings(p) == [p]


def test_construction_many_patterns_and_store_heuristic():
    rng = random.Random(3)
    pats = [f"{''.join(rng.choice('abcdefgh') for _ in range(rng.randint(3, 9)))}_{i}_"
            for i in range(30000)]
    a = AhoCorasick(pats)  # > 4096 code points: patterns are not stored, strings come from slices
    for p in pats[::997]:
        assert a.find_matches_as_strings(p) == [p]
    bp = [p.encode() for p in pats]
    b = BytesAhoCorasick(bp)
    for p in bp[::997]:
        assert [p[s:e] for (_, s, e) in b.find_matches_as_indexes(p)] == [p]


@FAST
@given(st.text(), st.text(min_size=1), st.text(), st.sampled_from([True, False, None]))
def test_str_prefix_pattern_suffix(prefix, pattern, suffix, store_patterns):
    hay = prefix + pattern + suffix
    a = AhoCorasick([pattern]) if store_patterns is None else AhoCorasick(
        [pattern], store_patterns=store_patterns)
    idx = a.find_matches_as_indexes(hay)
    assert {i for (i, _, _) in idx} == {0}
    assert {hay[s:e] for (_, s, e) in idx} == {pattern}
    assert set(a.find_matches_as_strings(hay)) == {pattern}


@FAST
@given(st.binary(), st.binary(min_size=1), st.binary())
def test_bytes_prefix_pattern_suffix(prefix, pattern, suffix):
    hay = prefix + pattern + suffix
    idx = BytesAhoCorasick([pattern]).find_matches_as_indexes(hay)
    assert {i for (i, _, _) in idx} == {0}
    assert {hay[s:e] for (_, s, e) in idx} == {pattern}


@FAST
@given(st.text(min_size=1), st.text(), st.sampled_from([True, False, None]))
def test_str_totally_random(pattern, hay, store_patterns):
    a = AhoCorasick([pattern]) if store_patterns is None else AhoCorasick(
        [pattern], store_patterns=store_patterns)
    idx = a.find_matches_as_indexes(hay)
    strs = a.find_matches_as_strings(hay)
    k = hay.find(pattern)
    if k == -1:
        assert idx == [] and strs == []
    else:
        assert idx[0][1] == k and hay[idx[0][1]:idx[0][2]] == pattern and strs[0] == pattern


@FAST
@given(st.binary(min_size=1), st.binary())
def test_bytes_totally_random(pattern, hay):
    idx = BytesAhoCorasick([pattern]).find_matches_as_indexes(hay)
    k = hay.find(pattern)
    if k == -1:
        assert idx == []
    else:
        assert idx[0][1] == k and hay[idx[0][1]:idx[0][2]] == pattern


@FAST
@given(st.lists(st.text(alphabet="abé☃🤦 ", min_size=1, max_size=5), min_size=1, max_size=12),
       st.text(alphabet="abé☃🤦 ", max_size=80), st.sampled_from([0, 1, 2]), st.booleans())
def test_str_all_kinds_vs_spec(pats, hay, mk, overlapping):
    a = AhoCorasick(pats, matchkind=KINDS[mk])
    if overlapping and mk != 0:
        with pytest.raises(ValueError):
            a.find_matches_as_indexes(hay, overlapping=True)
        return
    want = spec(pats, hay, mk, overlapping)
    assert a.find_matches_as_indexes(hay, overlapping=overlapping) == want
    assert a.find_matches_as_strings(hay, overlapping=overlapping) == [pats[i] for (i, _, _) in want]


# ------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("implementation", [None, Implementation.DFA, Implementation.ContiguousNFA])
def test_golden_reference_vectors(implementation):
    for c in load("reference_vectors.json"):
        pats, hay = c["patterns"], c["haystack"]
        a = AhoCorasick(pats, matchkind=KINDS[c["kind"]], implementation=implementation)
        b = BytesAhoCorasick([p.encode() for p in pats], matchkind=KINDS[c["kind"]],
                             implementation=implementation)
        if c.get("error"):
            with pytest.raises(ValueError):
                a.find_matches_as_indexes(hay, overlapping=c["overlapping"])
            with pytest.raises(ValueError):
                b.find_matches_as_indexes(hay.encode(), overlapping=c["overlapping"])
            continue
        idx = a.find_matches_as_indexes(hay, overlapping=c["overlapping"])
        assert [pats[i] for (i, _, _) in idx] == c["strings"], c["cite"]
        assert [hay[s:e] for (_, s, e) in idx] == c["strings"], c["cite"]
        assert a.find_matches_as_strings(hay, overlapping=c["overlapping"]) == c["strings"]
        if "indexes" in c:
            assert [list(m) for m in idx] == c["indexes"], c["cite"]
        hb = hay.encode()
        bidx = b.find_matches_as_indexes(hb, overlapping=c["overlapping"])
        assert [hb[s:e].decode() for (_, s, e) in bidx] == c["strings"], c["cite"]


def test_golden_leftmost_longest_genuine_crate():
    for c in load("ll_crate.json"):
        a = AhoCorasick(c["patterns"], matchkind=MatchKind.LeftmostLongest)
        got = [list(m) for m in a.find_matches_as_indexes(c["haystack"])]
        assert got == c["expected"], (c["patterns"], c["haystack"])


def test_golden_leftmost_longest_large_genuine_crate():
    import gen
    for c in load("ll_crate_large.json"):
        pats = list(dict.fromkeys(gen.gen_patterns(c["n_patterns_requested"], c["lo"], c["hi"],
                                                   gen.AZ_UNI, c["pattern_seed"])))
        hay = gen.gen_unicode_textlike(c["nchars"], c["haystack_seed"], pats,
                                       plant_every=c.get("plant_every", 512))
        got = AhoCorasick(pats, matchkind=MatchKind.LeftmostLongest).find_matches_as_indexes(hay)
        assert len(got) == c["count"]
        assert [list(m) for m in got[:16]] == c["head"]
        assert gen.canonical_sha256(got) == c["sha256"]


def test_golden_leftmost_first_re():
    for c in load("lf_re.json"):
        a = AhoCorasick(c["patterns"], matchkind=MatchKind.LeftmostFirst)
        got = [list(m) for m in a.find_matches_as_indexes(c["haystack"])]
        assert got == c["expected"], (c["patterns"], c["haystack"])


@pytest.mark.parametrize("implementation", [None, Implementation.DFA, Implementation.NoncontiguousNFA])
@pytest.mark.parametrize("case", load("kinds_large.json"),
                         ids=lambda c: f'{c["generator"]}{c["n_patterns"]}')
def test_golden_large_pattern_sets_all_kinds(case, implementation):
    """kinds_large.json through the HIP path (both scan kernels via the implementation hint):
    10k patterns with duplicates / nested pieces, every search mode, SHA-256 of the stream."""
    import gen
    pats, hay = gen.large_case_inputs(case)
    modes = [("standard", 0, False), ("overlapping", 0, True), ("leftmost_first", 1, False),
             ("leftmost_longest", 2, False)]
    for name, mk, ov in modes:
        want = case["results"][name]
        b = BytesAhoCorasick(pats, matchkind=KINDS[mk], implementation=implementation)
        got = b.find_matches_as_indexes(hay, overlapping=ov)
        assert len(got) == want["count"], name
        assert [list(m) for m in got[:16]] == want["head"], name
        assert gen.canonical_sha256(got) == want["sha256"], name
