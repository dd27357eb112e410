"""GPU parity: the HIP pipeline (through the C ABI) against the oracle
(oracle/ac_oracle.c) on seeded inputs, bit-exact.  Both scan kernels."""
import random

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu

capi = pytest.importorskip("ahocorasick_rs_amd.capi")

KERNELS = [capi.KERNEL_DFA_WALK, capi.KERNEL_PREFILTER]


def as_tuples(arr):
    return [(int(p), int(s), int(e)) for (p, s, e) in arr]


def check(pats, hay, mk, kernel, overlapping=False):
    a = capi.Automaton(pats, mk, kernel=kernel)
    got = as_tuples(a.find(hay, overlapping=overlapping))
    want = Oracle(pats, mk, KIND_DFA).find(hay, overlapping=overlapping)
    a.close()
    assert got == want, (pats[:8], bytes(hay[:80]), mk, kernel, overlapping, got[:5], want[:5])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_random_small(mk, kernel):
    rng = random.Random(1000 + mk)
    for it in range(60):
        alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
        pats = [bytes(rng.choice(alpha) for _ in range(rng.randint(1, 7)))
                for _ in range(rng.randint(1, 20))]
        hay = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 300)))
        check(pats, hay, mk, kernel)
        if mk == 0:
            check(pats, hay, 0, kernel, overlapping=True)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_medium_10k_patterns(mk, kernel):
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    a = capi.Automaton(pats, mk, kernel=kernel)
    o = Oracle(pats, mk, KIND_DFA)
    for hay in (gen.gen_uniform(1 << 20, gen.AZ, 12), gen.gen_textlike(1 << 20, 11, pats)):
        got = a.find(hay)
        want = o.find_raw(hay)
        assert len(got) == len(want)
        assert np.array_equal(np.stack([got["pattern"], got["start"], got["end"]], 1), want)
        if mk == 0:
            got = a.find(hay, overlapping=True)
            want = o.find_raw(hay, overlapping=True)
            assert np.array_equal(np.stack([got["pattern"], got["start"], got["end"]], 1), want)
    a.close()


def test_overlapping_error_and_empty():
    a = capi.Automaton([b"ab"], 2)
    with pytest.raises(ValueError):
        a.find(b"abab", overlapping=True)
    assert len(a.find(b"")) == 0
    a.close()
    z = capi.Automaton([], 0)
    assert len(z.find(b"anything")) == 0
    z.close()
    with pytest.raises(ValueError):
        capi.Automaton([b"x", b""], 0)
