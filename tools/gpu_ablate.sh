#!/bin/bash
# experiment: plain (cached) haystack loads in K1b -- duration and HBM fetch traffic
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ablate
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for ab in 0 16; do
  ACX_ABLATE=$ab timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$ab -o bench -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/t$ab.log 2>&1
  echo "== ablate $ab"; python /root/repo/tools/rocprof_summary.py $OUT/t$ab 2>/dev/null | grep -E "walk|k1b"
  ACX_ABLATE=$ab timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f$ab -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/f$ab.log 2>&1
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/f$ab/r_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][-28:]].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'k1b' in k or 'walk' in k or 'tile' in k: print(k, 'FETCH_SIZE avg', sum(v)/len(v))
PY
done
