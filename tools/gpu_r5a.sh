#!/bin/bash
# round 5, call A: early window gathers / larger sparse capacities -- parity of the variants, then same-box A/B
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r5a
mkdir -p $OUT
cd /root/repo
for v in early earlycap32; do
  ACX_LIB=/root/repo/variants/libacx_$v.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_$v.log 2>&1
  echo "pytest $v rc=$?"; tail -2 $OUT/pytest_$v.log
done
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
for rep in 1 2; do
  run T_tree_$rep tree ""
  run T_early_$rep early ""
  run T_cap32_$rep cap32 ""
  run T_earlycap32_$rep earlycap32 ""
  run cfg4_tree_$rep tree "--config cfg4"
  run cfg4_early_$rep early "--config cfg4"
  run cfg5_tree_$rep tree "--config cfg5"
  run cfg5_early_$rep early "--config cfg5"
done
run Z_tree tree "--dist Z"
run Z_early early "--dist Z"
cd /tmp
P="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-size --no-cold --no-secondary"
for v in tree early; do
  if [ "$v" = tree ]; then unset ACX_LIB; else export ACX_LIB=/root/repo/variants/libacx_$v.so; fi
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc_$v -o r -- $P > $OUT/pmc_tcc_$v.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc_cfg4_$v -o r -- $P --config cfg4 > $OUT/pmc_tcc_cfg4_$v.log 2>&1
done
unset ACX_LIB
cd /root/repo
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r5a/pmc_tcc_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k1b" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
