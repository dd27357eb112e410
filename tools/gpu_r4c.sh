#!/bin/bash
# round 4, call C: anchors (parity first), K0's polled result, RCCL behind the C ABI; same-box bench pairs
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4c
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu > $OUT/pytest_r4.log 2>&1
echo "round4 tests rc=$?"; tail -15 $OUT/pytest_r4.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_sparse_path.py tests/test_api_gpu.py tests/test_gpu_cfg1.py tests/test_gpu_batch.py -x -q -m gpu > $OUT/pytest_some.log 2>&1
echo "parity/configs/fullsize/sparse/api/cfg1/batch rc=$?"; tail -5 $OUT/pytest_some.log
run() { # tag, env..., (BARGS)
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
BARGS="--config cfg5" run cfg5 A=1
BARGS="--config cfg5" run cfg5_noanchors ACX_NO_ANCHORS=1
BARGS="--config mixedb" run mixedb A=1
BARGS="--config cfg3" run cfg3 A=1
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison.txt 2> $OUT/bench_comparison.err; head -12 $OUT/bench_comparison.txt
ACX_SMALL_SYNC=1 timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_sync.txt 2> $OUT/bench_comparison_sync.err; head -12 $OUT/bench_comparison_sync.txt
cd /tmp
rm -rf $OUT/trace_cfg5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg5 -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --config cfg5 > $OUT/trace_cfg5.log 2>&1
python /root/repo/tools/rocprof_summary.py $OUT/trace_cfg5 2>/dev/null | head -12
