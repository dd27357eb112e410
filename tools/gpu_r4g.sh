#!/bin/bash
# round 4, call G: the tile-ordered dense path after its two fixes (aggregated arrival counters, register sort)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4g
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "dense" > $OUT/pytest_a.log 2>&1
echo "dense tests rc=$?"; tail -12 $OUT/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_sparse_path.py tests/test_gpu_fuzz.py "tests/test_gpu_round3.py::test_dense_path_1gib_equals_sparse_path" -x -q -m gpu > $OUT/pytest_b.log 2>&1
echo "sparse seams / fuzz / dense 1 GiB rc=$?"; tail -6 $OUT/pytest_b.log
run() { # tag, env..., (BARGS)
  tag=$1; shift
  env "$@" timeout 600 python bench.py ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "cold", c["value_no_settle"])
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary"
BARGS="$Q --dist D" run D_tiles A=1
BARGS="$Q --dist D" run D_radix ACX_NO_DENSE_TILES=1
BARGS="$Q --config mixedx" run mixedx_tiles A=1
BARGS="$Q --config mixedx" run mixedx_radix ACX_NO_DENSE_TILES=1
cd /tmp
rm -rf $OUT/trace_D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_D -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist D > $OUT/trace_D.log 2>&1
python /root/repo/tools/rocprof_summary.py $OUT/trace_D 2>/dev/null | head -10
