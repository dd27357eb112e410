"""Pins the oracle (oracle/ac_oracle.c, the CPU restatement of the reference's
algorithm) against:
  * every known-answer vector of the reference's own tests / README,
  * the genuine aho-corasick 1.1.4 crate (LeftmostLongest, via fixtures
    captured from `tokenizers` by tests/golden/make_golden.py),
  * Python `re` alternation (LeftmostFirst fixtures),
  * the brute-force semantic spec (tests/spec.py) for all kinds.
CPU-only."""
import json
import os
import random

import pytest

import gen
from oracle_lib import KIND_DFA, KIND_NFA, Oracle, byte_to_code_point
from spec import spec

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name), encoding="utf-8") as f:
        return json.load(f)


@pytest.mark.parametrize("kind", [KIND_NFA, KIND_DFA])
@pytest.mark.parametrize("case", load("reference_vectors.json"),
                         ids=lambda c: c["cite"].replace("/", "_"))
def test_reference_known_answers(case, kind):
    pats = case["patterns"]
    hay = case["haystack"]
    o = Oracle([p.encode() for p in pats], case["kind"], kind)
    if case.get("error"):
        with pytest.raises(ValueError):
            o.find_str(hay, overlapping=case["overlapping"])
        return
    got = o.find_str(hay, overlapping=case["overlapping"])
    assert [pats[i] for (i, _, _) in got] == case["strings"]
    assert [hay[s:e] for (_, s, e) in got] == case["strings"]
    if "indexes" in case:
        assert [list(m) for m in got] == case["indexes"]
    # bytes twin: byte offsets slice the encoded haystack
    hb = hay.encode()
    gotb = o.find(hb, overlapping=case["overlapping"])
    assert [hb[s:e].decode() for (_, s, e) in gotb] == case["strings"]


@pytest.mark.parametrize("kind", [KIND_NFA, KIND_DFA])
def test_leftmost_longest_vs_genuine_crate(kind):
    for c in load("ll_crate.json"):
        o = Oracle([p.encode() for p in c["patterns"]], 2, kind)
        got = [list(m) for m in o.find_str(c["haystack"])]
        assert got == c["expected"], (c["patterns"], c["haystack"])


def test_leftmost_longest_large_vs_genuine_crate():
    for c in load("ll_crate_large.json"):
        pats = list(dict.fromkeys(gen.gen_patterns(c["n_patterns_requested"], c["lo"], c["hi"],
                                                   gen.AZ_UNI, c["pattern_seed"])))
        assert len(pats) == c["n_unique"]
        hay = gen.gen_unicode_textlike(c["nchars"], c["haystack_seed"], pats,
                                       plant_every=c.get("plant_every", 512))
        got = Oracle([p.encode() for p in pats], 2, KIND_DFA).find_str(hay)
        assert len(got) == c["count"]
        assert [list(m) for m in got[:16]] == c["head"]
        assert gen.canonical_sha256(got) == c["sha256"]


LARGE_MODES = [("standard", 0, False), ("overlapping", 0, True), ("leftmost_first", 1, False),
               ("leftmost_longest", 2, False)]


@pytest.mark.parametrize("kind", [KIND_NFA, KIND_DFA])
@pytest.mark.parametrize("case", load("kinds_large.json"),
                         ids=lambda c: f'{c["generator"]}{c["n_patterns"]}')
def test_large_pattern_sets_all_kinds(case, kind):
    """10k-pattern sets with duplicates and nested patterns: Standard / overlapping (brute-force
    spec), LeftmostFirst (`re` alternation), LeftmostLongest (spec) -- SHA-256 of the canonical
    stream.  Pins the duplicate tie-break (lowest index) at scale."""
    pats, hay = gen.large_case_inputs(case)
    assert len(pats) - len(set(pats)) == case["n_duplicates"] > 0
    for name, mk, ov in LARGE_MODES:
        want = case["results"][name]
        got = Oracle(pats, mk, kind).find(hay, overlapping=ov)
        assert len(got) == want["count"], name
        assert [list(m) for m in got[:16]] == want["head"], name
        assert gen.canonical_sha256(got) == want["sha256"], name


@pytest.mark.parametrize("kind", [KIND_NFA, KIND_DFA])
def test_leftmost_first_vs_re(kind):
    for c in load("lf_re.json"):
        o = Oracle([p.encode() for p in c["patterns"]], 1, kind)
        got = [list(m) for m in o.find_str(c["haystack"])]
        assert got == c["expected"], (c["patterns"], c["haystack"])


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_all_kinds_vs_bruteforce_spec(mk):
    rng = random.Random(100 + mk)
    for it in range(600):
        alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 6)))
                for _ in range(rng.randint(1, 14))]
        h = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 80)))
        want = spec(pats, h, mk)
        for kind in (KIND_NFA, KIND_DFA):
            assert Oracle(pats, mk, kind).find(h) == want
        if mk == 0:
            want = spec(pats, h, 0, overlapping=True)
            for kind in (KIND_NFA, KIND_DFA):
                assert Oracle(pats, 0, kind).find(h, overlapping=True) == want


def test_property_vectors_of_reference_tests():
    """tests/test_ac_bytes.py:133-161,175-189 restated with a seeded PRNG."""
    rng = random.Random(5)
    # construction_extensive: every pattern b"%b_%i_" is found as itself
    pats = [b"%b_%d_" % (bytes(rng.randrange(256) for _ in range(rng.randint(3, 9))), i)
            for i in range(3000)]
    o = Oracle(pats, 0, KIND_DFA)
    for p in pats[::37]:
        assert [p[s:e] for (_, s, e) in o.find(p)] == [p]
    # totally_random: first match start == haystack.find(pattern)
    for _ in range(500):
        pat = bytes(rng.randrange(4) for _ in range(rng.randint(1, 4)))
        hay = bytes(rng.randrange(4) for _ in range(rng.randint(0, 40)))
        got = Oracle([pat], 0, KIND_NFA).find(hay)
        idx = hay.find(pat)
        if idx == -1:
            assert got == []
        else:
            assert got[0][1] == idx and hay[got[0][1]:got[0][2]] == pat


def test_empty_and_zero_patterns():
    assert Oracle([], 0, KIND_DFA).find(b"anything") == []
    assert Oracle([b"a"], 0, KIND_DFA).find(b"") == []
    with pytest.raises(ValueError):
        Oracle([b""], 0, KIND_DFA)
    with pytest.raises(ValueError):
        Oracle([b"a"], 2, KIND_DFA).find(b"a", overlapping=True)


def test_byte_to_code_point():
    s = "a☃é🤦b"
    b = s.encode()
    m = byte_to_code_point(b)
    U = (1 << 64) - 1
    assert list(m) == [0, 1, U, U, 2, U, 3, U, U, U, 4, 5]
    assert list(byte_to_code_point(b"")) == [U]
