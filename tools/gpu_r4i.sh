#!/bin/bash
# round 4, call I: same-box pairs -- k_tile_main's launch bound on cfg5, the dense path's group size
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4i
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "dense or anchored or str_api" > $OUT/pytest_a.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/pytest_a.log
ACX_LIB=/root/repo/variants/libacx_g8.so timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k "dense" > $OUT/pytest_g8.log 2>&1
echo "tests (8-tile groups) rc=$?"; tail -4 $OUT/pytest_g8.log
run() { # tag, env..., (BARGS)
  tag=$1; shift
  env "$@" timeout 600 python bench.py ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "cold", c["value_no_settle"])
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary"
for i in 1 2; do
BARGS="$Q --config cfg5" run cfg5_bound$i A=1
BARGS="$Q --config cfg5" run cfg5_nobound$i ACX_LIB=/root/repo/variants/libacx_nobound.so
done
BARGS="$Q --dist D" run D_g4 A=1
BARGS="$Q --dist D" run D_g8 ACX_LIB=/root/repo/variants/libacx_g8.so
BARGS="$Q --config mixedx" run mixedx_g4 A=1
BARGS="$Q --config mixedx" run mixedx_g8 ACX_LIB=/root/repo/variants/libacx_g8.so
