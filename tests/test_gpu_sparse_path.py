"""GPU: the sparse output path (bucket slots + tile kernels) at its seams -- greedy chains
that cross tile boundaries, bucket / tile overflow into the dense (region + radix sort) path,
the hold-off after a dense call, and the chunked two-stream variant (ACX_CHUNKS)."""
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
TILE = 64 * 4096  # bytes of stream position per tile (TILE_BUCKETS * 4 KiB)


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


def check(a, o, hay, mk):
    for ov in ([False, True] if mk == 0 else [False]):
        assert np.array_equal(cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)), (mk, ov)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mk", [0, 1, 2])
def test_chains_across_bucket_and_tile_boundaries(mk, kernel):
    # runs of mutually overlapping occurrences laid over every kind of boundary: the greedy
    # chain of a tile then starts in the previous tile (lookback in k_tile_sort, chain
    # re-derivation from HBM in k_tile_resolve)
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
    for env in ({"ACX_CHUNKS": "3"}, {"ACX_NO_BUCKET": "1"}):
        out = "/tmp/acx_sparse_path_%s.npy" % "_".join(env)
        subprocess.run([sys.executable, "-c", code, out], check=True, env={**os.environ, **env}, timeout=600)
        assert np.array_equal(np.load(out), want), env
