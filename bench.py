#!/usr/bin/env python3
"""bench.py -- haystack GB/s of the MI355X-native Aho-Corasick hot path.

Metric (BASELINE.json): haystack GB/s (+ % of the HBM roofline), 10k-pattern DFA.

  N = 1   cfg2: 10 000 patterns (a-z, len 5-12, SplitMix64 seed 1), Implementation.DFA,
          ONE 1 GiB text-like synthetic haystack (seed 11, one pattern planted per KiB),
          MatchKind.Standard, non-overlapping, byte offsets (BytesAhoCorasick path).
  N > 1   cfg3: the same automaton on every GPU; a batch of 8 KiB haystacks cut from one
          global SplitMix64 stream, 131 072 haystacks (1 GiB) per GPU (weak scaling); no
          data-path collective; RCCL all-gather of the per-shard match counts only.

A "step" is one complete pass of the hot path over the rank's resident batch: scan kernel
(K1b prefilter) -> verification (k_walk_hits) -> tile kernels (sort, match-kind resolution,
scan, write) -> final (pattern, start, end) u64 triples in HBM, their count on the host
(+ the count all-gather for N > 1).  Inputs are resident in HBM before the timed region;
nothing is cached between steps.

One JSON line on rank 0; `roofline` is for the dominant kernel (K1) from HIP events recorded
on the library's stream inside the timed region; `cpu_baseline` is the oracle's C DFA loop
(a port of the reference's algorithm, 1 core) on the same haystack, timed on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
GIB = 1 << 30


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=GIB, help="haystack bytes per GPU")
    ap.add_argument("--dist", choices=["T", "U"], default="T",
                    help="T text-like (headline), U iid-uniform a-z")
    ap.add_argument("--kernel", choices=["auto", "dfa_walk", "prefilter"], default="auto")
    ap.add_argument("--workload", choices=["auto", "single", "batch"], default="auto",
                    help="auto: cfg2 single haystack at N=1, cfg3 batch of 8 KiB haystacks at N>1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-bytes", type=int, default=GIB)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import gen
    from ahocorasick_rs_amd import capi

    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    capi.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    # ---- automaton (cfg2 / cfg3)
    patterns = gen.gen_patterns(10000, 5, 12, gen.AZ, 1)
    impl = capi.IMPL_DFA
    kern = {"auto": None, "dfa_walk": capi.KERNEL_DFA_WALK,
            "prefilter": capi.KERNEL_PREFILTER}[args.kernel]
    ac = capi.Automaton(patterns, capi.MATCH_STANDARD, impl, kernel=kern)
    info = ac.info

    # ---- synthetic haystack, generated in HBM by the library (bit-exact twin of tests/gen.py)
    nbytes = args.bytes
    hay = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    kind, seed = (1, 11) if args.dist == "T" else (0, 12)
    if world > 1 or args.workload == "batch":
        seed = 13  # cfg3: ONE global stream cut into 8 KiB haystacks, sharded by rank
    torch.cuda.synchronize()
    ac.generate(hay.data_ptr(), nbytes, kind, seed, stream_offset=rank * nbytes)
    batch = world > 1 if args.workload == "auto" else args.workload == "batch"
    uniform_len = 8192 if batch else 0
    n_hay = nbytes // uniform_len if batch else 0
    if batch and nbytes % uniform_len:
        raise SystemExit("--bytes must be a multiple of 8192 for the batch workload")

    counts_local = torch.zeros(1, dtype=torch.int64, device=dev)
    counts_all = torch.zeros(world, dtype=torch.int64, device=dev)

    def step() -> int:
        r = ac.find_device(hay.data_ptr(), nbytes, n_hay=n_hay, uniform_len=uniform_len)
        n = r.count
        if dist is not None:  # C1: per-shard match counts -> global output offsets
            counts_local.fill_(n)
            dist.all_gather_into_tensor(counts_all, counts_local)
        r.free()
        return n

    for _ in range(args.warmup):
        n_matches = step()
    ac.profile_enable(True)
    ac.profile_read(reset=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_matches = step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = ac.profile_read(reset=True)
    ac.profile_enable(False)

    total_matches = n_matches
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        total_matches = int(counts_all.sum().item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * nbytes * args.steps / elapsed / 1e9
        scan_ms = prof.scan_ms / max(prof.scan_launches, 1)
        algo_bytes = nbytes + 24 * n_matches  # 1 B read / haystack byte + 24 B / match
        achieved = algo_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        out = {
            "metric": "haystack GB/s, 10k-pattern DFA",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": ("cfg3: 10k patterns a-z len 5-12 (seed 1), Implementation.DFA, batch of "
                             f"{n_hay} x 8 KiB haystacks per GPU from one SplitMix64 stream (seed 13), "
                             "MatchKind.Standard, sharded by haystack, RCCL all-gather of match counts")
                if batch else
                ("cfg2: 10k patterns a-z len 5-12 (seed 1), Implementation.DFA, one "
                 f"{nbytes / GIB:g} GiB {'text-like (T, seed 11)' if args.dist == 'T' else 'uniform a-z (U, seed 12)'}"
                 " bytes haystack, MatchKind.Standard, non-overlapping"),
                "bytes_per_gpu": nbytes,
                "n_patterns": len(patterns),
                "n_states": int(info.n_states),
                "dfa_table_bytes": int(info.table_bytes),
                "scan_kernel": capi.KERNEL_NAMES[info.kernel],
                "matches_per_gpu_step": int(n_matches),
                "matches_total": int(total_matches),
                "raw_occurrences_per_step": int(prof.raw_occurrences // max(prof.scan_launches, 1)),
                "prefix_hits_per_step": int(prof.prefix_hits // max(prof.scan_launches, 1)),
                "percent_of_hbm_roofline": round(100.0 * value / (HBM_PEAK_GBPS * world), 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k1b_prefilter" if info.kernel == capi.KERNEL_PREFILTER else "k1a_dfa_walk",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": measured_traffic(info, nbytes, args.dist, batch),
                "kernel_ms": round(scan_ms, 4),
                "algorithmic_bytes": int(algo_bytes),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(patterns, hay, min(args.cpu_sample_bytes, nbytes), n_matches, nbytes)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(info, nbytes, dist_name, batch):
    """HBM bytes per launch of the dominant kernel from the PMC passes of the SAME command
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction; produced by
    tools/collect_profiles.sh and committed under profiles/).  null when no measurement of this
    exact configuration is on file."""
    try:
        best = None
        for d in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
            f = os.path.join(ROOT, "profiles", d, "pmc_traffic.json")
            if os.path.exists(f):
                t = json.load(open(f))
                kern = "k1b_prefilter" if info.kernel == 2 else "k1a_dfa_walk"
                if (t.get("kernel") == kern and t.get("workload_bytes") == nbytes
                        and t.get("dist") == dist_name and not batch):
                    best = int(t["traffic_bytes"])
        return best
    except Exception:
        return None


def cpu_baseline(patterns, hay_t, sample_bytes, gpu_matches, nbytes):
    """The oracle's dense-DFA loop (C port of the reference's algorithm: class map, one
    dependent u32 load per byte, special-state range check), one core, on a prefix of the
    very same haystack.  A reported baseline, not the target."""
    try:
        from oracle_lib import KIND_DFA, Oracle
        host = hay_t[:sample_bytes].cpu().numpy()
        o = Oracle(patterns, 0, KIND_DFA)
        o.count(host[: 1 << 24])  # warm-up
        t0 = time.perf_counter()
        n = o.count(host)
        dt = time.perf_counter() - t0
        res = {"value": round(sample_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": 1,
               "kind": "port",
               "sample": f"first {sample_bytes / GIB:g} GiB of the same haystack, 1 pass, "
                         f"{dt:.1f} s, {n} matches"}
        if sample_bytes == nbytes:
            res["matches_equal_gpu"] = bool(n == gpu_matches)
        return res
    except Exception as e:  # the baseline must never take the GPU number down with it
        return {"value": None, "unit": "GB/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    main()
