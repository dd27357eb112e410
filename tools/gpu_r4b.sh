#!/bin/bash
# round 4, call B: the exact stage (parity first), then same-box bench lines with and without it
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4b
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu > $OUT/pytest_r4.log 2>&1
echo "round4 tests rc=$?"; tail -15 $OUT/pytest_r4.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_sparse_path.py -x -q -m gpu > $OUT/pytest_some.log 2>&1
echo "parity/configs/fullsize rc=$?"; tail -5 $OUT/pytest_some.log
run() { # tag, env..., -- bench args
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-target-size ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "cold", c["value_no_settle"])
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
BARGS="--config cfg2" run cfg2 A=1
BARGS="--config cfg2" run cfg2_exact ACX_K1B_EXACT=1
BARGS="--config cfg5" run cfg5 A=1
BARGS="--config cfg5" run cfg5_noexact ACX_K1B_EXACT=0
BARGS="--config mixedb" run mixedb A=1
BARGS="--config cfg2 --dist U" run cfg2U A=1
BARGS="--config cfg2 --dist U" run cfg2U_exact ACX_K1B_EXACT=1
cd /tmp
for cfg in cfg5; do
  rm -rf $OUT/trace_$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --config $cfg > $OUT/trace_$cfg.log 2>&1
  python /root/repo/tools/rocprof_summary.py $OUT/trace_$cfg 2>/dev/null | head -12
done
