"""bench.py plumbing that needs no GPU: the self-launcher (`--gpus N` without a launcher spawns N
ranks; here on a 2-rank gloo dry run, the same code path as the RCCL one minus the device
work), its refusal when fewer than N devices are visible, the exactness of the all-core CPU
baseline (ranges with max_len - 1 overlap + carry fix-up == one sequential pass), and the probe
for a genuine `ahocorasick_rs` wheel (this repository's own package of that name never counts)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import gen
from oracle_lib import KIND_DFA, Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run_bench(*argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_dry_run():
    p = run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    c = out["config"]
    assert c["world_size_observed"] == 2 and c["launcher"].startswith("self")
    assert len(c["per_rank_gbps"]) == 2 and c["matches_total"] == 1000 + 1001  # the count all-gather ran
    assert out["value"] > 0 and out["scaling"] == "weak"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a box WITHOUT enough GPUs")
def test_refuses_when_devices_are_missing():
    p = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and "HIP device" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("mk", [0, 1, 2])
def test_all_core_cpu_baseline_is_exact(mk):
    pats = gen.gen_patterns(2000, 3, 9, gen.AZ, 41) + [b"abab", b"bab", b"ababab"]
    hay = bytearray(gen.gen_textlike(1 << 18, 42, pats, plant_every=128).tobytes())
    for cut in range(1 << 14, 1 << 18, 1 << 14):  # matches straddling the cuts of a 16-way split
        hay[cut - 5:cut + 5] = b"ababababab"
    host = np.frombuffer(bytes(hay), dtype=np.uint8)
    o = Oracle(pats, mk, KIND_DFA)
    max_len = max(len(p) for p in pats)
    for threads in (1, 3, 16):
        got = bench.oracle_all_cores(o, host, max_len, threads, False)
        assert np.array_equal(got, o.find_raw(host)), (mk, threads)
        if mk == 0:
            got = bench.oracle_all_cores(o, host, max_len, threads, True)
            assert np.array_equal(got, o.find_raw(host, overlapping=True)), threads
    assert bench.sha256_stream(o.find_raw(host)) == gen.canonical_sha256(o.find(host))


def test_genuine_wheel_probe_ignores_this_repository():
    import ahocorasick_rs
    assert ahocorasick_rs.__acx_amd__ is True
    before = sys.modules["ahocorasick_rs"]
    assert bench.probe_genuine_wheel() is None  # no Rust wheel in this image; ours is not mistaken for one
    assert sys.modules["ahocorasick_rs"] is before and ROOT in [os.path.abspath(p) for p in sys.path]
