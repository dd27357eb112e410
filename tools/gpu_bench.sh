#!/bin/bash
# bench T/U + kernel trace + SQ instruction counters (no tests)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/quick
mkdir -p $OUT
cd /root/repo
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_T.json 2> $OUT/bench_T.err; cut -c1-200 $OUT/bench_T.json; grep -o '"roofline".*' $OUT/bench_T.json | cut -c1-230
timeout 200 python bench.py --steps 20 --warmup 3 --dist U --no-cpu-baseline > $OUT/bench_U.json 2> $OUT/bench_U.err; cut -c1-200 $OUT/bench_U.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_T.log 2>&1
cd /root/repo
python tools/rocprof_summary.py $OUT/trace_T 2>/dev/null | grep -E "walk|k1b|tile|fill|copy"
cd /tmp
rm -rf $OUT/pmc_sq
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/pmc_sq/r_counter_collection.csv')))
agg=collections.defaultdict(dict)
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][-28:]][r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items():
    if 'k1b' in k or 'walk' in k: print(k, v)
PY
