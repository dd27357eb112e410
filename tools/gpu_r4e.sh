#!/bin/bash
# round 4, call E: the whole GPU suite on the current tree, the default line, k_tile_main's SGPR cap as a same-box pair
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4e
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_api_gpu.py -x -q -m gpu > $OUT/pytest_a.log 2>&1
echo "round4+api rc=$?"; tail -6 $OUT/pytest_a.log
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round4.py --deselect tests/test_api_gpu.py > $OUT/pytest_b.log 2>&1
echo "rest rc=$?"; tail -6 $OUT/pytest_b.log
run() { # tag, env..., (BARGS)
  tag=$1; shift
  env "$@" timeout 600 python bench.py ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "cold", c["value_no_settle"])
    if "secondary" in c: print(json.dumps(c["secondary"]))
    if "target_8gib" in c: print(json.dumps(c["target_8gib"]))
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
Q="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary"
for i in 1 2; do
BARGS="$Q" run cfg2_cap$i A=1
BARGS="$Q" run cfg2_nocap$i ACX_LIB=/root/repo/variants/libacx_nosgprcap.so
done
BARGS="$Q --config cfg5" run cfg5_cap A=1
BARGS="$Q --config cfg5" run cfg5_nocap ACX_LIB=/root/repo/variants/libacx_nosgprcap.so
BARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" run default A=1
cd /tmp
for tag in cap nocap; do
  rm -rf $OUT/trace_$tag
  if [ $tag = nocap ]; then export ACX_LIB=/root/repo/variants/libacx_nosgprcap.so; else unset ACX_LIB; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --no-secondary > $OUT/trace_$tag.log 2>&1
  python /root/repo/tools/rocprof_summary.py $OUT/trace_$tag 2>/dev/null | head -6
done
