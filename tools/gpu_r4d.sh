#!/bin/bash
# round 4, call D: code points from the counts alone (str API), the default bench line with its secondary runs
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4d
mkdir -p $OUT
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_api_gpu.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_cfg1.py tests/test_gpu_fuzz.py tests/test_gpu_round3.py -x -q -m gpu > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -8 $OUT/pytest.log
run() { # tag, env..., (BARGS)
  tag=$1; shift
  env "$@" timeout 600 python bench.py ${BARGS} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$tag.json")); c = d["config"]; r = d["roofline"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "cold", c["value_no_settle"], "traffic", r["traffic"], r["traffic_source"])
    if "secondary" in c: print(json.dumps(c["secondary"]))
    if "target_8gib" in c: print(json.dumps(c["target_8gib"]))
    if "cpu_baseline" in d: print(json.dumps(d["cpu_baseline"])[:300])
except Exception as e:
    print("$tag failed", e); print(open("$OUT/bench_$tag.err").read()[-1500:])
PY
}
BARGS="--gpus 1 --steps 20 --warmup 5" run default A=1
BARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --config cfg5" run cfg5 A=1
BARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --config mixedb" run mixedb A=1
BARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-target-size --config cfg4" run cfg4 A=1
cd /tmp
rm -rf $OUT/trace_cfg5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --config cfg5 > $OUT/trace_cfg5.log 2>&1
python /root/repo/tools/rocprof_summary.py $OUT/trace_cfg5 2>/dev/null | head -10
