"""GPU tests added in round 3 (each one answers an item of VERDICT.md, round 2):

  * the verification of patterns whose bytes reach beyond what pinfo covers (ADVICE, high)
  * the Implementation hint never selects the slow scan kernel
  * device-resident haystacks through the façade (__dlpack__)            -- SURVEY §8 f2
  * one process, several devices: find_matches_as_indexes_batch(devices=[...]) / acx_find_batch_multi
  * RCCL executed once: bench.py as ONE rank under torch.distributed.run (nccl all-gather)
  * the dense (region) path at 1 GiB against the sparse path's result
  * cfg3 at its real per-GPU size (131 072 x 8 KiB), 2 % of the haystacks against the oracle
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

pytestmark = pytest.mark.gpu
capi = pytest.importorskip("ahocorasick_rs_amd.capi")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GIB = 1 << 30


def cols(a):
    return np.stack([a["pattern"], a["start"], a["end"]], 1) if len(a) else np.zeros((0, 3), np.uint64)


@pytest.mark.parametrize("kernel", [None, capi.KERNEL_PREFILTER, capi.KERNEL_DFA_WALK])
@pytest.mark.parametrize("min_len", [1, 2, 3, 8, 12])
@pytest.mark.parametrize("max_len", [14, 15, 16, 17, 18, 20, 21])
def test_long_tail_bytes_are_verified(kernel, min_len, max_len):
    """A pattern has bytes to compare in place when it is longer than what pinfo holds (filter_q2 + 12
    bytes) OR longer than the 16 haystack bytes that travel with a hit: near misses that differ only in
    the last bytes must not be reported.  (Round 2 gated the comparison on max_len > 16 only: 14..16-byte
    patterns behind a 1..3-byte prefix slipped through; round 3's first fix gated it on filter_q2 + 12
    only: 17..20-byte patterns behind an 8-byte prefix did -- found by the fuzzer.)"""
    short = b"abcdefghijklmnopqrstuvwxyz"[:min_len]
    long_p = b"abcdefghijklmnopqrstuvwxyz"[:max_len]
    pats = [short, long_p]
    near = [long_p[:-1] + b"#", long_p[:-2] + b"#" + long_p[-1:], long_p[:-3] + b"#" + long_p[-2:]]
    rng = np.random.default_rng(max_len * 7 + min_len)
    hay = bytearray(rng.integers(48, 58, 40000, dtype=np.uint8).tobytes())  # digits: no accidental match
    for k, p in enumerate(range(100, 39000, 131)):
        x = near[k % 3] if k % 4 else long_p
        hay[p:p + len(x)] = x
    hay = bytes(hay)
    for mk in (0, 1, 2):
        a = capi.Automaton(pats, mk, kernel=kernel)
        o = Oracle(pats, mk, KIND_DFA)
        for ov in ([False, True] if mk == 0 else [False]):
            got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
            assert np.array_equal(got, want), (mk, ov, min_len, max_len, kernel)
        a.close()


def test_implementation_hint_never_selects_the_slow_kernel():
    pats = gen.gen_patterns(2000, 5, 12, gen.AZ, 1)
    hay = gen.gen_textlike(1 << 20, 11, pats)
    want = None
    for impl in (capi.IMPL_AUTO, capi.IMPL_NONCONTIGUOUS_NFA, capi.IMPL_CONTIGUOUS_NFA, capi.IMPL_DFA):
        a = capi.Automaton(pats, 0, impl)
        assert capi.KERNEL_NAMES[a.info.kernel] == "prefilter", impl
        got = cols(a.find(hay))
        want = got if want is None else want
        assert np.array_equal(got, want)
        a.close()


_DLPACK_SCRIPT = r"""
import sys
import torch  # first: one process holds ONE HIP runtime, and torch must be the one to load it (package docs)
sys.path[:0] = [sys.argv[1], sys.argv[2]]
import gen
import ahocorasick_rs as ref_api
pats = gen.gen_patterns(3000, 5, 12, gen.AZ, 1)
host = gen.gen_textlike(8 << 20, 11, pats)
ac = ref_api.BytesAhoCorasick(pats, matchkind=ref_api.MatchKind.LeftmostLongest)
want = ac.find_matches_as_indexes(host.tobytes())
t = torch.from_numpy(host).to("cuda:0")
assert ac.find_matches_as_indexes(t) == want and len(want) > 1000
assert ac.find_matches_as_indexes(t[1234567:]) == ac.find_matches_as_indexes(host[1234567:].tobytes())
# host tensors go through the same protocol; small ones take K0
assert ac.find_matches_as_indexes(torch.from_numpy(host[:5000])) == ac.find_matches_as_indexes(host[:5000].tobytes())
std = ref_api.BytesAhoCorasick(pats)
assert std.find_matches_as_indexes(t, overlapping=True) == std.find_matches_as_indexes(host.tobytes(), overlapping=True)
for bad, exc in ((t.reshape(2, -1), TypeError), (t.to(torch.int8), BufferError), (t[::2], TypeError)):
    try:
        ac.find_matches_as_indexes(bad)
    except exc:
        pass
    else:
        raise AssertionError(f"no {exc.__name__}")
print("DLPACK_OK")
"""


def test_facade_accepts_device_resident_tensor():
    """BytesAhoCorasick.find_matches_as_indexes(tensor in HBM) == the bytes call, no H2D copy of the
    haystack (the tensor is searched where it lies).  In a process of its own: torch has to be the
    first to load the HIP runtime, whatever ran before in this session."""
    pytest.importorskip("torch")
    p = subprocess.run([sys.executable, "-c", _DLPACK_SCRIPT, ROOT, os.path.join(ROOT, "tests")],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "DLPACK_OK" in p.stdout, p.stderr[-3000:]


def test_batch_over_devices_single_process():
    """find_matches_as_indexes_batch(devices=[...]): one process, one host thread per device.  One
    GPU on the test box: devices=[0] is the plain call, and acx_find_batch_multi over several handles
    of device 0 exercises the split / merge (contiguous ranges, counts combined on the host)."""
    import ahocorasick_rs as ref_api
    pats = gen.gen_patterns(3000, 3, 9, gen.AZ, 21)
    hay = gen.gen_textlike(257 * 3000, 13, pats)
    hays = [hay[i * 3000:(i + 1) * 3000].tobytes() for i in range(257)] + [b"", b"x", pats[0]]
    ac = ref_api.BytesAhoCorasick(pats)
    want = [ac.find_matches_as_indexes(h) for h in hays]
    assert ac.find_matches_as_indexes_batch(hays) == want
    assert ac.find_matches_as_indexes_batch(hays, devices=[0]) == want
    assert ac.find_matches_as_indexes_batch(hays, overlapping=False, devices=(0,)) == want
    with pytest.raises(ValueError):
        ac.find_matches_as_indexes_batch(hays, devices=[])
    with pytest.raises(ValueError):
        ac.find_matches_as_indexes_batch(hays, devices=[capi.device_count() + 3])
    sp = [p.decode() for p in pats]
    sac = ref_api.AhoCorasick(sp, matchkind=ref_api.MatchKind.LeftmostFirst)
    shays = [h.decode() for h in hays] + ["é☃ " + sp[5] + " 🤦" + sp[6]]
    assert sac.find_matches_as_indexes_batch(shays, devices=[0]) == [sac.find_matches_as_indexes(h) for h in shays]
    # several handles (replicas on device 0): the split / merge of acx_find_batch_multi
    a = capi.Automaton(pats, 0)
    reps = [a.replicate(0) for _ in range(3)]
    m1, c1 = a.find_batch(hays)
    for k in (1, 2, 3):
        m, c = a.find_batch_multi(reps[:k], hays)
        assert np.array_equal(c, c1) and np.array_equal(cols(m), cols(m1)), k
    m, c = a.find_batch_multi(reps, hays[:2], overlapping=True)  # fewer haystacks than handles
    m2, c2 = a.find_batch(hays[:2], overlapping=True)
    assert np.array_equal(c, c2) and np.array_equal(cols(m), cols(m2))
    for r in reps:
        r.close()
    a.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_one_rank_under_torch_distributed_run():
    """RCCL init + the count all-gather executed on the MI355X: bench.py as the single rank of a
    torch.distributed.run launch (backend nccl = RCCL)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
           "--steps", "2", "--warmup", "1", "--bytes", str(64 << 20), "--no-cpu-baseline", "--no-target-size"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    c = d["config"]
    assert c["launcher"] == "torch.distributed.run" and c["world_size_observed"] == 1
    assert c["collective"]["backend"] == "nccl" and c["collective"]["all_gathers"] >= 3
    assert d["n_gpus"] == 1 and c["matches_total"] == c["matches_per_gpu_step"] > 0


def test_dense_path_1gib_equals_sparse_path():
    """cfg2-T at 1 GiB forced onto the dense (region) path (ACX_NO_BUCKET=1: hit regions -> k_walk_hits
    -> radix sort -> resolve) equals the sparse path's stream, which test_baseline_size_1gib_bit_exact
    pins against the oracle element by element."""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(GIB)
    a.generate(buf.ptr, GIB, 1, 11)
    r = a.find_device(buf.ptr, GIB)
    sparse = cols(r.matches())
    r.free()
    os.environ["ACX_NO_BUCKET"] = "1"
    try:
        r = a.find_device(buf.ptr, GIB)
        dense = cols(r.matches())
        r.free()
        r = a.find_device(buf.ptr, GIB, overlapping=True)
        dense_ov = cols(r.matches())
        r.free()
    finally:
        del os.environ["ACX_NO_BUCKET"]
    r = a.find_device(buf.ptr, GIB, overlapping=True)
    sparse_ov = cols(r.matches())
    r.free()
    assert len(sparse) == 1094465  # (the count BENCH_r02.json / the oracle report for this input)
    assert np.array_equal(dense, sparse)
    assert np.array_equal(dense_ov, sparse_ov)
    a.close()


def test_cfg3_full_per_gpu_size_counts_and_samples():
    """cfg3 at its real per-GPU size: 131 072 haystacks of 8 KiB in one call.  sum(counts) == n,
    every match lies inside its haystack, and 2 % of the haystacks (every 50th) equal the oracle."""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    n_hay, L = 131072, 8192
    a = capi.Automaton(pats, 0, capi.IMPL_DFA)
    buf = capi.DeviceBuffer(n_hay * L)
    a.generate(buf.ptr, n_hay * L, 1, 13)
    r = a.find_device(buf.ptr, n_hay * L, n_hay=n_hay, uniform_len=L)
    m, counts = cols(r.matches()), r.counts().astype(np.int64)
    r.free()
    assert int(counts.sum()) == len(m) > n_hay  # (one planted pattern per KiB: ~8 per haystack)
    assert np.all(m[:, 2] <= L) and np.all(m[:, 1] < m[:, 2])
    starts = np.concatenate([[0], np.cumsum(counts)])
    host = buf.download()
    o = Oracle(pats, 0, KIND_DFA)
    for h in range(0, n_hay, 50):
        want = o.find_raw(host[h * L:(h + 1) * L])
        assert np.array_equal(m[starts[h]:starts[h + 1]], want), h
    a.close()


@pytest.mark.parametrize("short", [[b"qz", b"~"], [b"ab", b"x"], [b"b", b"abcd"]], ids=["rare", "frequent", "readme"])
def test_mixed_length_sets_leave_the_prefilter_for_the_failureless_walk(short):
    """A dictionary with a 1- or 2-byte pattern in it.  Round 3: K1b could not
    take it, the set ran the failureless walk (k1a_scan marks every triple below a short pattern, k1a_walk
    starts those positions at the root).  Round 4: K1b takes it after all (the short patterns through its
    side test, tests/test_gpu_round4.py) -- the failureless walk is still what an explicit dfa_walk runs.
    Rare short patterns stay on the sparse path, frequent ones (b"x": every 26th letter) end on the dense
    path -- every kind equals the oracle either way.
    ([b"b", b"abcd"] is the vector of /root/reference/README.md:106-108.)"""
    pats = gen.gen_patterns(10000, 5, 12, gen.AZ, 1) + short
    hay = gen.gen_textlike(4 << 20, 11, pats[:10000]).tobytes()
    for mk in (0, 1, 2):
        o = Oracle(pats, mk, KIND_DFA)
        for kernel, name in ((None, "prefilter"), (capi.KERNEL_DFA_WALK, "dfa_walk")):
            a = capi.Automaton(pats, mk, kernel=kernel)
            assert capi.KERNEL_NAMES[a.info.kernel] == name
            for ov in ([False, True] if mk == 0 else [False]):
                got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
                assert np.array_equal(got, want), (mk, ov, short, name)
            a.close()


def test_fuzz_case_long_tail_behind_an_eight_byte_prefix_dense_output():
    """The case tools/gpu_fuzz.py (seed 424242) reported against round 3's first fix of the tail
    gate: a 15- and a 20-byte pattern (filter_q2 = 8, pinfo covers 20 bytes, the carried window 16),
    planted every few bytes so that truncated copies abound and the call takes the dense path: the
    copies that differ from the 20-byte pattern in bytes 16..19 were reported."""
    pats = [b"usrtgsyauibwlnq", b"pffthapvlcxzctugopqc"]
    rng = np.random.default_rng(164)
    hay = bytearray(rng.choice(np.frombuffer(gen.AZ + b" ", dtype=np.uint8), 42722).tobytes())
    p = 0
    while p < len(hay) - 32:
        x = pats[int(rng.integers(0, 2))]
        hay[p:p + len(x)] = x
        p += int(rng.choice([9, 13, 17, 22]))
    hay = bytes(hay)
    for kernel in (None, capi.KERNEL_PREFILTER, capi.KERNEL_DFA_WALK):
        a = capi.Automaton(pats, 0, kernel=kernel)
        o = Oracle(pats, 0, KIND_DFA)
        for ov in (False, True):
            got, want = cols(a.find(hay, overlapping=ov)), o.find_raw(hay, overlapping=ov)
            assert len(want) > 300 and np.array_equal(got, want), (kernel, ov, len(got), len(want))
        a.close()
