"""Multi-GPU batches: one process per GPU, haystacks sharded by index.

The hot path shards naturally (independent haystacks), so there is NO data-path
collective: every rank scans its own contiguous range of haystacks with its own
replica of the automaton.  The only exchange is C1 of SURVEY.md §2: an
all-gather of the per-rank match counts (world x 8 bytes, latency-bound over
xGMI), from which every rank derives the global output offsets of its matches.

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
