#!/bin/bash
# round 4, call J: cfg5 tuning knobs as same-box lines (host-side table choices only: no kernel differs)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4j
mkdir -p $OUT
cd /root/repo
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
Q="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary --config cfg5"
BARGS="$Q" run cfg5_default A=1
BARGS="$Q" run cfg5_load4 ACX_PTAB_INV_LOAD=4
BARGS="$Q" run cfg5_load16 ACX_PTAB_INV_LOAD=16
BARGS="$Q" run cfg5_gain4 ACX_SHIFT_GAIN=4
BARGS="$Q" run cfg5_gain64 ACX_SHIFT_GAIN=64
BARGS="$Q" run cfg5_default2 A=1
