#!/bin/bash
# GPU check: tests (fail fast, most basic first), then short benches.  Every command
# bounded by `timeout`.  usage: tools/gpu_check.sh [tests|bench|all]
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/check
mkdir -p $OUT
cd /root/repo
WHAT=${1:-all}
if [ "$WHAT" = tests ] || [ "$WHAT" = all ]; then
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py \
      tests/test_gpu_compressed.py tests/test_gpu_cfg1.py tests/test_gpu_configs.py tests/test_api_gpu.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -25 $OUT/pytest.log
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  for d in T U; do
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dist $d > $OUT/bench_$d.json 2> $OUT/bench_$d.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$d.json")); c = d["config"]; r = d["roofline"]
    print("$d", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel_ms"], "ms frac", r["frac"], "matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "occ", c["raw_occurrences_per_step"])
except Exception as e:
    print("$d failed", e); print(open("$OUT/bench_$d.err").read()[-1500:])
PY
  done
fi
if [ "$WHAT" = trace ] || [ "$WHAT" = all ]; then
  cd /tmp
  rm -rf $OUT/trace_T
  TAG=${TRACE_TAG:-T}
  rm -rf $OUT/trace_$TAG
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o bench -- python /root/repo/bench.py --steps ${TRACE_STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/trace_$TAG.log 2>&1
  cd /root/repo
  python tools/rocprof_summary.py $OUT/trace_$TAG 2>/dev/null | head -16
fi
if [ "$WHAT" = pmc ]; then
  cd /tmp
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $OUT/pmc_$tag
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/pmc_$tag.log 2>&1
  done
  cd /root/repo
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_*/r_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")[:24]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith(("k1b", "k_tile", "k1a", "k_walk")):
        print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())})
PY
fi
if [ "$WHAT" = cfgs ]; then
  for cfg in cfg4 cfg4b cfg5 cfg3; do
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --config $cfg ${BENCH_ARGS:-} > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$cfg.json")); c = d["config"]; r = d["roofline"]
    print("$cfg", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel_ms"], "ms", r["kernel"], "matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "occ", c["raw_occurrences_per_step"])
except Exception as e:
    print("$cfg failed", e); print(open("$OUT/bench_$cfg.err").read()[-1500:])
PY
  done
fi
if [ "$WHAT" = full ]; then
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cat $OUT/bench_default.json | cut -c1-3000; tail -3 $OUT/bench_default.err
  timeout 600 python bench.py --host --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_host.json 2> $OUT/bench_host.err; cut -c1-600 $OUT/bench_host.json; tail -3 $OUT/bench_host.err
  for m in 1 2; do ACX_STAGE=$m timeout 600 python bench.py --host --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_host_stage$m.json 2> $OUT/bench_host_stage$m.err; echo "ACX_STAGE=$m"; cut -c1-200 $OUT/bench_host_stage$m.json; done
  timeout 900 python bench.py --bytes 8589934592 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_8g.json 2> $OUT/bench_8g.err; cut -c1-400 $OUT/bench_8g.json; tail -3 $OUT/bench_8g.err
fi
if [ "$WHAT" = k1a ]; then
  for d in T U; do
    ACX_WALK_STATS=${WALK_STATS:-} timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dist $d --kernel dfa_walk > $OUT/bench_k1a_$d.json 2> $OUT/bench_k1a_$d.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_k1a_$d.json")); c = d["config"]; r = d["roofline"]
    print("K1a $d", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel_ms"], "ms", r["kernel"], "matches", c["matches_total"])
except Exception as e:
    print("$d failed", e); print(open("$OUT/bench_k1a_$d.err").read()[-1500:])
PY
    grep "acx:" $OUT/bench_k1a_$d.err | tail -1
  done
fi
