#!/bin/bash
# round 4, call H: quick check of the block-scan version of k_dense_main + the slimmed bench loop
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4h
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu > $OUT/pytest_a.log 2>&1
echo "round4 tests rc=$?"; tail -5 $OUT/pytest_a.log
run() { # tag, env..., (BARGS)
  tag=$1; shift
  env "$@" timeout 600 python bench.py ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "cold", c["value_no_settle"])
    if "secondary" in c: print(json.dumps(c["secondary"]))
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary"
BARGS="$Q --dist D" run D_tiles A=1
BARGS="$Q --config mixedx" run mixedx_tiles A=1
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-target-size" run default A=1
