"""GPU: automata whose dense transition table is not kept (n_states * stride * 4 above
ACX_DENSE_LIMIT -- very large pattern sets).  Every kernel that walks the automaton then
steps the compressed form (trie edges + failure links, the classic Aho-Corasick transition);
the results must not change.  The tests force the compressed form with ACX_DENSE_LIMIT=0."""
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


@pytest.fixture
def compressed(monkeypatch):
    monkeypatch.setenv("ACX_DENSE_LIMIT", "0")
    yield
    monkeypatch.delenv("ACX_DENSE_LIMIT", raising=False)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_random_sets_without_dense_table(compressed, mk, kernel):
    rng = random.Random(1000 + mk)
    for it in range(12):
        alpha = [b"ab", b"abc", b"abcdefgh", bytes(range(256))][it % 4]
        pats = list({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 9))) for _ in range(rng.randint(1, 40))})
        n = [0, 1, 63, 300, 5000, 70000, 300000][it % 7]
        hay = bytes(rng.choice(alpha) for _ in range(n))
        a = capi.Automaton(pats, mk, kernel=kernel)
        assert a.info.table_bytes == 0
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            assert np.array_equal(cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)), (it, mk, ov)
        a.close()


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_large_dictionary_without_dense_table(compressed, mk):
    # SURVEY's text-like workload at 4 MiB: the prefilter path (which never reads the table) and,
    # for the second automaton, the walk over the compressed form
    pats = gen.gen_patterns(10000, 3, 12, gen.AZ, 5)
    hay = gen.gen_textlike(4 << 20, 6, pats, 4096).tobytes()
    o = Oracle(pats, mk, KIND_DFA)
    want = o.find_raw(hay, overlapping=False)
    for kernel in KERNELS:
        a = capi.Automaton(pats, mk, kernel=kernel)
        assert a.info.table_bytes == 0
        assert np.array_equal(cols(a.find(hay)), want), kernel
        # dense results (every position matches): the region path walks the automaton too
        dense_hay = b"".join(pats[:2000]) * 3
        assert np.array_equal(cols(a.find(dense_hay)), o.find_raw(dense_hay, overlapping=False)), kernel
        a.close()
