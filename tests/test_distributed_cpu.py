"""N > 1 path on CPU: world_size-2 gloo processes, sharding + count all-gather.
The scan itself needs a GPU, so each rank's shard is matched by the ORACLE here
(test infrastructure standing in for the device); what is under test is the
product's sharding / offset logic (ahocorasick_rs_amd/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleAutomaton:
    """find_batch() with the product's signature, computed by the oracle."""

    def __init__(self, pats, mk):
        from oracle_lib import KIND_DFA, Oracle
        self.o = Oracle(pats, mk, KIND_DFA)

    def find_batch(self, hays, overlapping=False, codepoints=False):
        per = [self.o.find_raw(h, overlapping) for h in hays]
        counts = np.array([len(p) for p in per], dtype=np.uint64)
        allm = np.concatenate(per) if per else np.zeros((0, 3), np.uint64)
        return allm, counts


def _worker(rank, world, port, ret):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib.util
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location(
        "acx_distributed", os.path.join(os.path.dirname(HERE), "ahocorasick_rs_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    import gen
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pats = gen.gen_patterns(300, 3, 8, gen.AZ, 21)
    hay = gen.gen_textlike(101 * 512, 13, pats)  # ONE stream cut into 101 haystacks: shard-count independent
    hays = [hay[i * 512:(i + 1) * 512].tobytes() for i in range(101)]
    res = D.find_batch_sharded(_OracleAutomaton(pats, 0), hays)
    ret[rank] = (res["lo"], res["hi"], res["matches"].tolist(), res["counts"].tolist(),
                 res["rank_counts"], res["global_offset"], res["global_total"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_equals_unsharded(world):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    sys.path.insert(0, HERE)
    import gen
    pats = gen.gen_patterns(300, 3, 8, gen.AZ, 21)
    hay = gen.gen_textlike(101 * 512, 13, pats)
    hays = [hay[i * 512:(i + 1) * 512].tobytes() for i in range(101)]
    whole, whole_counts = _OracleAutomaton(pats, 0).find_batch(hays)
    # shards are contiguous, ordered, and cover everything; the union in rank order is the whole
    los = [ret[r][0] for r in range(world)]
    his = [ret[r][1] for r in range(world)]
    assert los[0] == 0 and his[-1] == 101 and all(his[r] == los[r + 1] for r in range(world - 1))
    cat = sum((ret[r][2] for r in range(world)), [])
    assert cat == whole.tolist()
    assert sum((ret[r][3] for r in range(world)), []) == whole_counts.tolist()
    counts = [len(ret[r][2]) for r in range(world)]
    for r in range(world):
        assert ret[r][4] == counts                      # all-gathered counts agree on every rank
        assert ret[r][5] == sum(counts[:r])             # global offset of this shard's matches
        assert ret[r][6] == len(whole)                  # global total
        # the shard's matches sit at [offset, offset + count) of the global list
        assert whole.tolist()[ret[r][5]:ret[r][5] + counts[r]] == ret[r][2]


def test_shard_range_properties():
    sys.path.insert(0, os.path.dirname(HERE))
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "acx_distributed", os.path.join(os.path.dirname(HERE), "ahocorasick_rs_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    for n in (0, 1, 7, 8, 1048576):
        for w in (1, 2, 3, 8):
            rs = [D.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1
    assert D.exclusive_offsets([3, 0, 5]) == [0, 3, 3]
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


# ---------------------------------------------------------------------------
# ONE haystack cut into byte ranges (second row)
# ---------------------------------------------------------------------------
MATCH_DTYPE = np.dtype([("pattern", "<u8"), ("start", "<u8"), ("end", "<u8")])


class _OracleSingle:
    """find() / max_pattern_len with the product's (capi.Automaton) signature, from the oracle."""

    def __init__(self, pats, mk):
        from oracle_lib import KIND_DFA, Oracle
        self.o = Oracle(pats, mk, KIND_DFA)
        self.max_pattern_len = max(len(p) for p in pats)
        self.calls = 0

    def find(self, hay, overlapping=False, codepoints=False):
        self.calls += 1
        raw = self.o.find_raw(bytes(hay), overlapping)
        out = np.empty(len(raw), dtype=MATCH_DTYPE)
        out["pattern"], out["start"], out["end"] = raw[:, 0], raw[:, 1], raw[:, 2]
        return out


def _load_distributed():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "acx_distributed", os.path.join(os.path.dirname(HERE), "ahocorasick_rs_amd", "distributed.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)
    return D


def _single_case(mk):
    """Patterns that overlap one another heavily + a haystack whose matches straddle every cut
    of a 2- and 3-way split (so the carry exchange has to do real work)."""
    sys.path.insert(0, HERE)
    import gen
    pats = [b"abab", b"bab", b"ababab", b"cc", b"ccc", b"abcabcabcabc"] + gen.gen_patterns(40, 2, 6, b"abc", 31)
    hay = bytearray(gen.gen_uniform(6000, b"abcx", 32).tobytes())
    for cut in (2000, 3000, 4000):
        hay[cut - 7:cut + 7] = b"abababababcccc"
    return pats, bytes(hay)


def _worker_single(rank, world, port, mk, overlapping, ret):
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    D = _load_distributed()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pats, hay = _single_case(mk)
    a = _OracleSingle(pats, mk)
    res = D.find_single_sharded(a, hay, overlapping=overlapping)
    m = res["matches"]
    ret[rank] = (res["lo"], res["hi"], np.stack([m["pattern"], m["start"], m["end"]], 1).tolist(),
                 res["rank_counts"], res["global_offset"], res["global_total"], res["rounds"], a.calls)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mk,overlapping", [(0, False), (0, True), (1, False), (2, False)])
def test_single_haystack_sharded_equals_whole(world, mk, overlapping):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_single, args=(world, port, mk, overlapping, ret), nprocs=world, join=True)
    pats, hay = _single_case(mk)
    from oracle_lib import KIND_DFA, Oracle
    whole = Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping).tolist()
    cat = sum((ret[r][2] for r in range(world)), [])
    assert cat == whole
    counts = [len(ret[r][2]) for r in range(world)]
    for r in range(world):
        assert ret[r][3] == counts and ret[r][4] == sum(counts[:r]) and ret[r][5] == len(whole)
        if not overlapping:
            assert 1 <= ret[r][6] <= world  # carry rounds
    # the same answer from the sequential simulation (what the GPU tests use on one device)
    D = _load_distributed()
    sim = D.simulate_single_sharded(_OracleSingle(pats, mk), hay, world, overlapping)
    assert [np.stack([m["pattern"], m["start"], m["end"]], 1).tolist() for m in sim] == \
        [ret[r][2] for r in range(world)]


def test_single_haystack_ranges_shorter_than_a_pattern():
    # a match that swallows whole ranges: the carry passes through ranks that report nothing
    D = _load_distributed()
    pats = [b"abcdefghijklmnop", b"gh", b"p"]
    hay = b"xxabcdefghijklmnopxx"
    from oracle_lib import KIND_DFA, Oracle
    for mk in (0, 1, 2):
        for ov in ([False, True] if mk == 0 else [False]):
            whole = Oracle(pats, mk, KIND_DFA).find_raw(hay, ov).tolist()
            for world in (1, 2, 5, 7):
                sim = D.simulate_single_sharded(_OracleSingle(pats, mk), hay, world, ov)
                cat = sum((np.stack([m["pattern"], m["start"], m["end"]], 1).tolist() for m in sim), [])
                assert cat == whole, (mk, ov, world)


class _OracleDevice(_OracleSingle):
    """The same with find_device(): 'device pointers' are offsets into a host buffer the stub holds, so the
    device-resident form of the sharded search (distributed.DeviceHaystack) runs here without a GPU."""

    BASE = 0x7000_0000

    def __init__(self, pats, mk, hay):
        super().__init__(pats, mk)
        self.hay, self.freed, self.scanned = bytes(hay), 0, 0

    def find_device(self, ptr, nbytes, *, overlapping=False, **_):
        off = ptr - self.BASE
        assert 0 <= off and off + nbytes <= len(self.hay)
        self.scanned += nbytes
        got, me = self.find(self.hay[off:off + nbytes], overlapping), self

        class R:
            def matches(self):
                return got

            def free(self):
                me.freed += 1
        return R()


@pytest.mark.parametrize("mk,overlapping", [(0, False), (0, True), (1, False), (2, False)])
def test_single_haystack_sharded_device_resident_form(mk, overlapping):
    D = _load_distributed()
    pats, hay = _single_case(mk)
    from oracle_lib import KIND_DFA, Oracle
    whole = Oracle(pats, mk, KIND_DFA).find_raw(hay, overlapping).tolist()
    for world in (1, 2, 3, 8):
        a = _OracleDevice(pats, mk, hay)
        sim = D.simulate_single_sharded(a, D.DeviceHaystack(a.BASE, len(hay)), world, overlapping)
        cat = sum((np.stack([m["pattern"], m["start"], m["end"]], 1).tolist() for m in sim), [])
        assert cat == whole, (mk, overlapping, world)
        assert a.freed == a.calls >= world  # every device result released


def test_resume_stops_at_the_first_common_match():
    """The re-search after a changed carry scans windows until it reports a match the first pass reported
    too, then reuses the first pass's tail (it used to rescan the whole range)."""
    D = _load_distributed()
    import random
    rng = random.Random(11)
    for it in range(200):
        pats = [bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(1, 30))]
        mk = rng.choice([0, 1, 2])
        hay = bytes(rng.choice(b"abcx") for _ in range(rng.randint(50, 3000)))
        a = _OracleSingle(pats, mk)
        m = max(a.max_pattern_len - 1, 0)
        n = len(hay)
        lo = rng.randrange(0, n // 2)
        hi = rng.randrange(lo + 1, n + 1)
        last = hi == n or rng.random() < 0.2
        hi = n if last else hi
        c1 = rng.randrange(lo, min(hi, lo + rng.choice([1, 3, 10, 200])) + 1)
        old = D._local_greedy(a, memoryview(hay), lo, hi, m, last)
        want = D._local_greedy(a, memoryview(hay), c1, hi, m, last)
        got, scanned = D._resume_greedy(a, memoryview(hay), c1, hi, m, last, old, window=rng.choice([16, 64, 1 << 16]))
        for k in ("pattern", "start", "end"):
            assert got[k].tolist() == want[k].tolist(), (it, k)
    # and it is cheap: a 1 MB range whose carry moves by a few bytes is not rescanned
    sys.path.insert(0, HERE)
    import gen
    pats = [b"abab", b"bab", b"cc"] + gen.gen_patterns(20, 3, 6, b"abc", 5)
    hay = gen.gen_uniform(1 << 20, b"abcxyzuvw", 6).tobytes()
    a = _OracleSingle(pats, 1)
    m = a.max_pattern_len - 1
    old = D._local_greedy(a, memoryview(hay), 0, len(hay), m, True)
    want = D._local_greedy(a, memoryview(hay), 3, len(hay), m, True)
    got, scanned = D._resume_greedy(a, memoryview(hay), 3, len(hay), m, True, old, window=4096)
    assert got.tolist() == want.tolist()
    assert scanned <= 4 * (4096 + m), scanned
