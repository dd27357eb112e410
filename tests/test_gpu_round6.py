"""GPU, round 6: the forms of the sparse path's post stage and of k_dense_main against the oracle.

* narrow staged words (32-bit words, 128 threads, 8-byte reported occurrences) are what cfg2-shaped sets run by default --
  ACX_MAIN_WIDE=1 puts the same inputs through the wide-word form;
* the WIDE FORM (64 occurrences per bucket, 4 096 matches per group): forced on ordinary inputs (ACX_FORCE_WIDE=1: every
  instantiation -- anchors, code points, narrow and wide words -- sees sparse AND dense inputs), and taken by a context by
  itself when its groups mostly give up on the narrow stage (a match every 256 bytes), left again on a sparse input;
* k_dense_main's compact form against its full form (ACX_NO_DENSE_COMPACT=1), and a call whose groups do not fit the compact
  stage (more than 1 024 staged occurrences per group of four tiles) repeated in the full form.
The environment switches are read once per process: subprocesses."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
HERE = os.path.dirname(os.path.abspath(__file__))
from test_gpu_staged import SCRIPT  # noqa: E402  (all kinds + overlapping, short patterns, code points + anchors, a dense stretch, a batch)


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


@pytest.mark.parametrize("env", [{"ACX_FORCE_WIDE": "1"}, {"ACX_MAIN_WIDE": "1"}, {"ACX_FORCE_WIDE": "1", "ACX_MAIN_WIDE": "1"},
                                 {"ACX_NO_DENSE_COMPACT": "1"}])
def test_forms_of_the_post_stage_forced(env):
    r = subprocess.run([sys.executable, "-c", SCRIPT], env={**os.environ, **env}, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]


def planted(n, step, pats, seed):
    hay = gen.gen_textlike(n, 11, pats).copy()
    rng = gen.SplitMix64(seed)
    for k in range(0, n - 32, step):
        p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
        hay[k:k + len(p)] = p
    return hay.tobytes()


def test_a_context_takes_the_wide_form_and_leaves_it_again():
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    dense = planted(8 << 20, 256, pats, 3)   # ~20 occurrences per 4 KiB bucket: the narrow stage's buckets overflow
    denser = planted(8 << 20, 128, pats, 4)  # ~36: the wide form's 64 slots hold them
    sparse = gen.gen_textlike(8 << 20, 12, pats).tobytes()
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk, capi.IMPL_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            a.path_stats(reset=True)
            for h in (dense, denser, dense[5:]):
                assert np.array_equal(cols(a.find(h, overlapping=ov)), o.find_raw(h, overlapping=ov)), (mk, ov)
            st = a.path_stats()
            assert st["wide_redone"] >= 1 and st["dense_tiles"] == st["dense_radix"] == 0, st
            # a sparse input: still the wide form for this call, the narrow one from the next on -- never the dense path
            for _ in range(2):
                assert np.array_equal(cols(a.find(sparse, overlapping=ov)), o.find_raw(sparse, overlapping=ov))
            st = a.path_stats()
            assert st["sparse"] == 2 and st["wide_redone"] == 0 and st["hot_calls"] == 0, st
        a.close()


def test_the_compact_dense_main_hands_a_crowded_call_to_the_full_form():
    # a pattern every 16 bytes: 256+ occurrences per tile, five staged tiles of a dense group hold more than the compact
    # stage's 1 024 words -- the kernel is repeated in its full form; every 64 bytes: the compact form takes it
    pats = gen.gen_patterns(3000, 5, 9, gen.AZ, 2)
    for step in (16, 64):
        hay = planted(6 << 20, step, pats, step)
        for mk, ov in ((0, False), (0, True), (2, False)):
            a = capi.Automaton(pats, mk, capi.IMPL_DFA)
            want = Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping=ov)
            for _ in range(2):  # (the second call: the context's hold on the dense path / on the full form)
                got = cols(a.find(hay, overlapping=ov))
                assert got.shape == want.shape and np.array_equal(got, want), (step, mk, ov)
            st = a.path_stats()
            assert st["dense_tiles"] >= 1, st
            a.close()


def test_the_hot_pipeline_queued_ahead_of_the_knowledge_that_it_is_needed():
    # a context whose last call had a few hot groups queues the hot pipeline right behind the tile kernels, its grids for a
    # bound of hot groups, their number read on the device (kernels.hip: hot_groups_here).  The calls of one handle in turn:
    # one hot region (the first: the old way; the next ones: speculative), more regions than the bound (the kernels return,
    # the host runs the pipeline the old way), none at all (the speculative kernels return at once), one again
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    n = 24 << 20  # 96 groups of 256 KiB

    def with_regions(starts, seed):
        hay = gen.gen_textlike(n, seed, pats).copy()
        rng = gen.SplitMix64(seed)
        for s0 in starts:
            for k in range(s0, s0 + (48 << 10), 32):
                p = np.frombuffer(pats[rng.next() % len(pats)], dtype=np.uint8)
                hay[k:k + len(p)] = p
        return hay.tobytes()

    inputs = [with_regions([5 << 20], 21), with_regions([9 << 20], 22), with_regions([(9 << 20) + 100_000], 23),
              with_regions([k << 20 for k in (1, 3, 6, 8, 11, 14, 17, 20)], 24), gen.gen_textlike(n, 25, pats).tobytes(),
              gen.gen_textlike(n, 26, pats).tobytes(), with_regions([2 << 20], 27), with_regions([(2 << 20) + 70_000, 20 << 20], 28)]
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        a = capi.Automaton(pats, mk, capi.IMPL_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            a.path_stats(reset=True)
            for i, h in enumerate(inputs):
                got, want = cols(a.find(h, overlapping=ov)), o.find_raw(h, overlapping=ov)
                assert got.shape == want.shape and np.array_equal(got, want), (mk, ov, i)
            st = a.path_stats()
            assert st["hot_calls"] == 6 and st["sparse"] == 2 and st["dense_tiles"] == st["dense_radix"] == 0, st
        a.close()


FRESH_SCRIPT = r"""
import os, sys
sys.path.insert(0, os.path.join(os.environ["ACX_ROOT"], "tests")); sys.path.insert(0, os.environ["ACX_ROOT"])
import numpy as np, gen
from ahocorasick_rs_amd import capi
def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)
pats = gen.gen_patterns(3000, 3, 9, gen.AZ, 21)
hay = gen.gen_textlike(257 * 3000, 13, pats)
hays = [hay[i * 3000:(i + 1) * 3000].tobytes() for i in range(257)] + [b"", b"x", pats[0]]
other = capi.Automaton(pats, 0)
bad = 0
for it in range(60):
    for h in hays[:40]:
        other.find(h)  # (launched K0 calls on another handle: more streams, the device kept busy)
    a = capi.Automaton(pats, 0)
    reps = [a.replicate(0) for _ in range(3)]
    m1, c1 = a.find_batch(hays)
    for k in (1, 2, 3):
        m, c = a.find_batch_multi(reps[:k], hays)
        bad += not (np.array_equal(c, c1) and np.array_equal(cols(m), cols(m1)))
    for r in reps:
        r.close()
    a.close()
print("BAD", bad)
"""


def test_a_fresh_contexts_first_call_beside_a_busy_device():
    """The clearing of a fresh workspace (control blocks, overflow counters, supergroup words) used to be hipMemset calls on
    the NULL stream, which the contexts' non-blocking streams do not wait for: with another thread keeping the device busy
    a fresh handle's first batch ran its scan before the counters were cleared and lost overflow hits (10-17 wrong batches
    in 100 rounds of this loop; found by the full suite's order of tests in round 6).  Everything is queued on the context's
    own stream now."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", FRESH_SCRIPT], env={**os.environ, "ACX_ROOT": root, "ACX_NO_RESIDENT": "1"},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BAD 0" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
