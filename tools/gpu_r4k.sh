#!/bin/bash
# round 4, call K: k_tile_main with one thread per staged occurrence, and the fused write (look-back), against the
# previous commit's library on the same box (ACX_LIB) -- tests first
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4k
mkdir -p $OUT
cd /root/repo
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_sparse_path.py tests/test_gpu_round3.py tests/test_gpu_batch.py -x -q > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -5 $OUT/pytest.log
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
PREV=/root/repo/ahocorasick_rs_amd/libacx_hip_prev.so
Q="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary"
for cfg in "" "--config cfg5" "--config mixed" "--config cfg4" "--dist U"; do
  t=$(echo "$cfg" | tr -d ' -' ); t=${t:-T}
  BARGS="$Q $cfg" run ${t}_prev ACX_LIB=$PREV
  BARGS="$Q $cfg" run ${t}_phases ACX_NO_FUSED_WRITE=1
  BARGS="$Q $cfg" run ${t}_fused A=1
done
BARGS="$Q" run T_prev2 ACX_LIB=$PREV
BARGS="$Q" run T_fused2 A=1
