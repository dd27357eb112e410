"""GPU parity at BASELINE.json cfg1's shape: the reference benchmark's name list
(/root/reference/benchmarks/test_comparison.py:16-18 -- 4 244 lower-cased names, ~5 % of them
duplicates) over 1 MB of ASCII prose.  names.txt does not travel to the GPU box, so the set is
the seeded stand-in of SURVEY.md §8d (tests/gen.py names_like(4244, seed 6)); what matters is
the duplicate tie-break (lowest pattern index) on the general K1b -> candidate list -> tile
kernel path, on the plain DFA walk, on K0 and on the batch path -- all against the oracle."""
import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
KERNELS = [capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER]


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


@pytest.fixture(scope="module")
def cfg1():
    names = gen.names_like(4244, 6)
    pats = [p.encode() for p in names]
    assert len(pats) - len(set(pats)) > 200  # the duplicates are there
    return names, pats, gen.names_haystack(names, 1_000_000)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_cfg1_single_1mb_haystack(cfg1, mk, kernel):
    _, pats, hay = cfg1
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    want = o.find_raw(hay)
    got = cols(a.find(hay))
    assert len(want) > 300 and np.array_equal(got, want)
    # the planted names include duplicated ones: the lower index of a pair is reported
    dup_hits = [int(p) for p in got[:, 0] if pats.count(pats[int(p)]) > 1]
    assert dup_hits and all(pats.index(pats[p]) == p for p in dup_hits)
    if mk == 0:
        wo = o.find_raw(hay, overlapping=True)
        go = cols(a.find(hay, overlapping=True))
        assert len(wo) > len(want) and np.array_equal(go, wo)  # both copies of a duplicate, id order
    a.close()


@pytest.mark.parametrize("kernel", [None] + KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_cfg1_batch_of_lines(cfg1, mk, kernel):
    """The reference's benchmark shape: ~600-character lines, one call each (K0 when no kernel
    is forced) and all of them in one device pass."""
    names, pats, _ = cfg1
    lines = [l.encode() for l in gen.names_lines(names, 1500, every=3)]
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    for ov in ([False, True] if mk == 0 else [False]):
        m, counts = a.find_batch(lines, overlapping=ov)
        want = [o.find_raw(l, overlapping=ov) for l in lines]
        assert counts.tolist() == [len(w) for w in want]
        assert np.array_equal(cols(m), np.concatenate(want))
        for l, w in list(zip(lines, want))[:150]:  # per-call path
            assert np.array_equal(cols(a.find(l, overlapping=ov)), w)
    a.close()


def test_cfg1_str_api(cfg1):
    import ahocorasick_rs_amd as ac
    names, pats, hay = cfg1
    s = hay.decode("ascii")
    for mk, kind in ((0, ac.MatchKind.Standard), (1, ac.MatchKind.LeftmostFirst),
                     (2, ac.MatchKind.LeftmostLongest)):
        a = ac.AhoCorasick(names, matchkind=kind)
        want = Oracle(pats, mk, KIND_DFA).find(hay)
        assert a.find_matches_as_indexes(s) == want
        assert a.find_matches_as_strings(s) == [names[i] for (i, _, _) in want]
