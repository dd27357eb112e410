"""Multi-GPU: one process per GPU.

Batches (cfg3): haystacks sharded by index.  The hot path shards naturally (independent
haystacks), so there is NO data-path collective: every rank scans its own contiguous range
of haystacks with its own replica of the automaton.  The only exchange is C1 of SURVEY.md §2:
an all-gather of the per-rank match counts (world x 8 bytes, latency-bound over xGMI), from
which every rank derives the global output offsets of its matches.

One large haystack (second row): byte ranges with an overlap of
max_pattern_len - 1 bytes; the only exchange is the greedy's carry -- where the previous
rank's last match ends -- as an all-gather of one integer per rank (find_single_sharded).

Works with any torch.distributed backend: `nccl` (= RCCL on ROCm) on GPUs,
`gloo` in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` owned by `rank`; sizes differ by at most one and
    the concatenation over ranks is 0..n_items in order (so global order = rank order)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def exclusive_offsets(counts: Sequence[int]) -> List[int]:
    out, run = [], 0
    for c in counts:
        out.append(run)
        run += int(c)
    return out


def gather_match_counts(local_count: int, group=None, device=None) -> Tuple[List[int], int, int]:
    """All-gather of the per-rank match counts.
    Returns (counts per rank, this rank's global offset, global total)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return [int(local_count)], 0, int(local_count)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
    every = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(every, mine, group=group)
    counts = [int(x) for x in every.cpu().tolist()]
    return counts, exclusive_offsets(counts)[rank], sum(counts)


def capi_comm_from_env(device: int, timeout_s: float = 120.0):
    """A `capi.Comm` (RCCL behind the C ABI: no torch.distributed) for a process started by any launcher
    that sets RANK / WORLD_SIZE / MASTER_PORT (torch.distributed.run does).  The 128-byte RCCL id goes from
    rank 0 to the others through a file next to the rendezvous port -- the host's own means, as
    include/acx.h puts it; an MPI or socket host would carry it its own way."""
    import os
    import tempfile
    import time
    from . import capi
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # the name is made of what EVERY rank of a job shares by construction -- the rendezvous address and port, the elastic
    # agent's run id and restart count, or an explicit ACX_COMM_NONCE -- so ranks started by hand in separate shells, or each
    # behind its own wrapper (numactl, bash -c), find the same file; only when none of those is set does the parent's pid
    # stand in (the ranks of one launcher).  Rank 0 removes what it finds before it creates the id, and what it wrote once
    # every rank holds the communicator.
    shared = [os.environ.get(k) for k in ("ACX_COMM_NONCE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT")]
    tag = "_".join(v for v in shared if v) or f"ppid{os.getppid()}"
    tag = "".join(ch if ch.isalnum() or ch in "._-" else "-" for ch in tag)
    path = os.path.join(tempfile.gettempdir(), f"acx_comm_{tag}.id")
    if rank == 0:
        for stale in (path, path + ".tmp"):
            try:
                os.unlink(stale)
            except FileNotFoundError:
                pass
        uid = capi.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
    else:
        t0 = time.time()
        uid = b""
        while len(uid) != 128:  # (the id is written to a temporary name and renamed: a short read means another writer)
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank 0 never published the RCCL id at {path}")
            try:
                with open(path, "rb") as f:
                    uid = f.read()
            except FileNotFoundError:
                uid = b""
            if len(uid) != 128:
                time.sleep(0.01)
    comm = capi.Comm.init_rank(uid, world, rank, device)  # (collective: returns when every rank has joined)
    if rank == 0:
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
    return comm


def gather_match_counts_capi(comm, local_count: int, rank: int) -> Tuple[List[int], int, int]:
    """gather_match_counts through the C ABI's communicator (acx_comm_allgather_counts + acx_output_offsets)."""
    from . import capi
    counts = comm.allgather_counts([int(local_count)])
    off = capi.output_offsets(counts)
    return counts, off[rank], off[-1]


def find_batch_sharded(automaton, haystacks: Sequence[bytes], overlapping: bool = False,
                       codepoints: bool = False, group=None):
    """Scan this rank's shard of `haystacks` (every rank passes the same list, or at least
    a list of the same length) and exchange the counts.

    `automaton` is any object with find_batch(list, overlapping=, codepoints=) ->
    (matches, counts), e.g. ahocorasick_rs_amd.capi.Automaton built on this rank's GPU.
    Returns dict(lo, hi, matches, counts, rank_counts, global_offset, global_total)."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(len(haystacks), rank, world)
    matches, counts = automaton.find_batch(list(haystacks[lo:hi]), overlapping=overlapping,
                                           codepoints=codepoints)
    rank_counts, off, total = gather_match_counts(len(matches), group=group)
    return {"lo": lo, "hi": hi, "matches": matches, "counts": counts,
            "rank_counts": rank_counts, "global_offset": off, "global_total": total}


# ---------------------------------------------------------------------------
# one large haystack cut into byte ranges
# ---------------------------------------------------------------------------
class DeviceHaystack:
    """A haystack resident in HBM on THIS rank's device: a device pointer and its length (e.g.
    `torch.Tensor.data_ptr()`, `capi.DeviceBuffer.ptr`).  find_single_sharded / simulate_single_sharded take it in
    place of a bytes-like object: every range is then scanned where it lies (automaton.find_device), only the
    match records travel to the host.  (Round 3 had the host form only: Python slices of a host bytes object.)"""

    def __init__(self, ptr: int, nbytes: int):
        self.ptr, self.nbytes = int(ptr), int(nbytes)

    def __len__(self) -> int:
        return self.nbytes


def _as_hay(haystack):
    return haystack if isinstance(haystack, DeviceHaystack) else memoryview(haystack).cast("B")


def _scan(automaton, hay, a: int, b: int, overlapping: bool):
    """The matches of hay[a:b], offsets local to a."""
    if isinstance(hay, DeviceHaystack):
        r = automaton.find_device(hay.ptr + a, max(b - a, 0), overlapping=overlapping)
        try:
            return r.matches()
        finally:
            r.free()
    return automaton.find(hay[a:b], overlapping=overlapping)


def _shift(matches, by: int):
    if by and len(matches):
        matches = matches.copy()
        matches["start"] += by
        matches["end"] += by
    return matches


def _local_overlapping(automaton, hay, lo: int, hi: int, m: int):
    """All occurrences that END in (lo, hi] (rank 0: in (0, hi]); global offsets.  An occurrence
    that ends in the range starts at most m = max_pattern_len - 1 bytes before it."""
    w = max(0, lo - m)
    got = _shift(_scan(automaton, hay, w, hi, True), w)
    return got[got["end"] > lo] if lo > 0 else got


def _local_greedy(automaton, hay, carry: int, hi: int, m: int, last: bool):
    """The non-overlapping matches that START in [carry, hi), global offsets, given that the
    global iteration resumes at `carry` (the end of the last match that starts before this
    range, or the range's own start).  A match that starts before hi ends at most m bytes
    after it, and no occurrence the truncated window hides can beat one that it shows."""
    n = len(hay)
    got = _shift(_scan(automaton, hay, carry, n if last else min(n, hi + m), False), carry)
    return got if last else got[got["start"] < hi]


RESYNC_WINDOW = 1 << 16


def _resume_greedy(automaton, hay, carry: int, hi: int, m: int, last: bool, old, window: int = RESYNC_WINDOW):
    """_local_greedy(carry) when `old` = _local_greedy(some other carry) of the same range is at hand: the
    non-overlapping iteration is a function of the position it resumes at, so as soon as the new iteration
    reports a match the old one reported too, the two coincide from there on.  Scans windows from `carry`
    until that happens (one window unless matches pile up) and splices the old tail on -- instead of
    rescanning the whole range (round 3).  Returns (matches, bytes scanned)."""
    import numpy as np
    n = len(hay)
    end = n if last else min(n, hi + m)
    stop = n if last else hi
    old = old[old["start"] >= carry] if len(old) else old
    parts, scanned, c = [], 0, carry
    while c < stop:
        w_end = min(end, c + window + m)
        got = _shift(_scan(automaton, hay, c, w_end, False), c)
        scanned += w_end - c
        if w_end < end:  # (a match that starts in the window's last m bytes may be cut short: leave it to the next window)
            got = got[got["start"] < c + window]
        got = got[got["start"] < stop] if not last else got
        if len(got) and len(old):
            # the first new match that is also an old one (both are sorted by start, a start occurs once in either)
            j = np.searchsorted(old["start"], got["start"])
            jc = np.minimum(j, len(old) - 1)
            same = (j < len(old)) & (old["start"][jc] == got["start"]) & (old["end"][jc] == got["end"]) & \
                   (old["pattern"][jc] == got["pattern"])
            if same.any():
                i = int(np.argmax(same))
                parts.append(got[:i + 1])
                parts.append(old[int(j[i]) + 1:])
                return np.concatenate(parts), scanned
        parts.append(got)
        c = max(c + window, int(got["end"][-1]) if len(got) else 0) if w_end < end else stop
        old = old[old["start"] >= c] if len(old) else old
    return (np.concatenate(parts) if parts else old[:0]), scanned


def _carry_out(matches, hi: int, carry_in: int) -> int:
    """Where the global iteration stands when it leaves a range that ends at hi."""
    return max(hi, carry_in, int(matches["end"][-1]) if len(matches) else 0)


def simulate_single_sharded(automaton, haystack, world: int, overlapping: bool = False):
    """What find_single_sharded returns on every rank of a `world`-rank job, computed
    sequentially in one process (tests, single-GPU validation).  List of per-rank match arrays;
    their concatenation equals automaton.find(haystack, overlapping)."""
    hay = _as_hay(haystack)
    m = max(int(automaton.max_pattern_len) - 1, 0)
    out, carry = [], 0
    for rank in range(world):
        lo, hi = shard_range(len(hay), rank, world)
        if overlapping:
            out.append(_local_overlapping(automaton, hay, lo, hi, m))
            continue
        carry = max(carry, lo)
        got = _local_greedy(automaton, hay, carry, hi, m, rank == world - 1)
        out.append(got)
        carry = _carry_out(got, hi, carry)
    return out


def find_single_sharded(automaton, haystack, overlapping: bool = False, group=None):
    """This rank's part of automaton.find(haystack, overlapping) when ONE haystack (every rank
    passes the same bytes-like object) is cut into `world` contiguous byte ranges.

    overlapping: a rank reports the occurrences that END in its range (window = range plus
    max_pattern_len - 1 bytes to the left); no exchange beyond the counts.
    non-overlapping (any match kind): a rank reports the matches that START in its range.  The
    only coupling is the carry: a match of the previous rank that ends inside this range moves
    the point the iteration resumes at.  Every rank first scans speculatively from its range
    start; the carries are all-gathered (one integer per rank); a rank whose true carry
    differs rescans from it; repeat until nothing changes (one round unless a match straddles
    a cut, at most `world` rounds).  Offsets are global byte offsets.

    `automaton`: find(bytes-like, overlapping=) -> structured array (pattern, start, end) and
    max_pattern_len, e.g. ahocorasick_rs_amd.capi.Automaton.  `haystack`: a bytes-like object, or a
    DeviceHaystack (the haystack -- at least this rank's range plus max_pattern_len - 1 bytes either side --
    resident on this rank's device: the ranges are scanned in HBM, automaton.find_device).
    Returns dict(lo, hi, matches, rank_counts, global_offset, global_total, rounds)."""
    import torch
    import torch.distributed as dist
    hay = _as_hay(haystack)
    on = dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    world = dist.get_world_size(group) if on else 1
    lo, hi = shard_range(len(hay), rank, world)
    m = max(int(automaton.max_pattern_len) - 1, 0)
    rounds = 0
    if overlapping:
        got = _local_overlapping(automaton, hay, lo, hi, m)
    else:
        device = None
        if on:
            device = torch.device("cuda", torch.cuda.current_device()) \
                if dist.get_backend(group) == "nccl" else torch.device("cpu")
        carry, last = lo, rank == world - 1
        got = _local_greedy(automaton, hay, carry, hi, m, last)
        while on and world > 1:
            rounds += 1
            mine = torch.tensor([_carry_out(got, hi, carry)], dtype=torch.int64, device=device)
            every = torch.zeros(world, dtype=torch.int64, device=device)
            dist.all_gather_into_tensor(every, mine, group=group)
            outs = [int(x) for x in every.cpu().tolist()]
            want = lo if rank == 0 else max(lo, outs[rank - 1])
            changed = want != carry
            if changed:  # resume at the true carry and stop as soon as the iteration is back on the old track
                carry = want
                got, _ = _resume_greedy(automaton, hay, carry, hi, m, last, got)
            flag = torch.tensor([1 if changed else 0], dtype=torch.int64, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
            if int(flag.item()) == 0:
                break
    rank_counts, off, total = gather_match_counts(len(got), group=group)
    return {"lo": lo, "hi": hi, "matches": got, "rank_counts": rank_counts,
            "global_offset": off, "global_total": total, "rounds": rounds}
