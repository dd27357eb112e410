#!/bin/bash
# experiment: K1b duration + FETCH_SIZE for the current build
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ablate
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o r -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/f.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/f/r_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][-28:]].append(float(r['Counter_Value']))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open('$OUT/f/r_kernel_trace.csv')) if 'k1b' in r['Kernel_Name']]
print('k1b us: min %.1f median %.1f' % (min(d), sorted(d)[len(d)//2]), {k:sum(v)/len(v) for k,v in agg.items() if 'k1b' in k})
PY
cd /root/repo
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
