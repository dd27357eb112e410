#!/bin/bash
# round 5, call B: the hot pipeline (parity + H1 / H100 / density / D), early windows fixed, LDS alignment, cfg4 issue counters
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r5b
mkdir -p $OUT
cd /root/repo
timeout 60 tools/ubench_lds_align.bin > $OUT/lds_align.txt 2>&1; head -20 $OUT/lds_align.txt
timeout 900 python -m pytest tests/test_gpu_hot.py -x -q -m gpu > $OUT/pytest_hot.log 2>&1
echo "pytest hot rc=$?"; tail -15 $OUT/pytest_hot.log
timeout 900 python -m pytest tests/test_gpu_sparse_path.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_batch.py -x -q -m gpu > $OUT/pytest_paths.log 2>&1
echo "pytest paths rc=$?"; tail -8 $OUT/pytest_paths.log
ACX_LIB=/root/repo/variants/libacx_early.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_early.log 2>&1
echo "pytest early rc=$?"; tail -3 $OUT/pytest_early.log
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold --no-secondary"
run() { # name lib args
  if [ "$2" = tree ]; then unset ACX_LIB; else export ACX_LIB=/root/repo/variants/libacx_$2.so; fi
  timeout 300 python bench.py $Q $3 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/$1.json")); r = d["roofline"]
    print("$1:", d["ms_per_step"], "ms/step", d["value"], "GB/s  K1", r["kernel_ms"], "ms  matches", d["config"]["matches_total"])
except Exception as e:
    print("$1 failed", e); print(open("$OUT/$1.err").read()[-600:])
PY
  unset ACX_LIB
}
run T_tree tree ""
run H1 tree "--dist H1"
run H100 tree "--dist H100"
run T_tree2 tree ""
for n in 1024 512 256 128 64 32; do run P$n tree "--dist P$n --steps 10 --warmup 3"; done
run D tree "--dist D --steps 10 --warmup 3"
run cfg4_tree tree "--config cfg4"
run cfg4_early early "--config cfg4"
run cfg5_tree tree "--config cfg5"
run cfg5_early early "--config cfg5"
cd /tmp
P="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-size --no-cold --no-secondary"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq_cfg4 -o r -- $P --config cfg4 > $OUT/pmc_sq_cfg4.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds_cfg4 -o r -- $P --config cfg4 > $OUT/pmc_lds_cfg4.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_H1 -o bench -- $P --dist H1 --steps 5 > $OUT/trace_H1.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r5b/pmc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1b" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: round(sum(v) / len(v)) for k, v in agg.items()})
for f in glob.glob("gpurun_out/r5b/trace_H1/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
