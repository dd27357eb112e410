#!/bin/bash
# K1b ablations: 1 = level 1 only, 2 = no hit push, 5 = loads only (timing + VALU count; results wrong by construction)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ablate
rm -rf $OUT; mkdir -p $OUT
cd /tmp
for ab in 0 1 2 5; do
  ACX_ABLATE=$ab timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/p$ab -o r -- python /root/repo/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/p$ab.log 2>&1
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open('$OUT/p$ab/r_counter_collection.csv')))
agg=collections.defaultdict(dict)
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][-28:]][r['Counter_Name']]=float(r['Counter_Value'])
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open('$OUT/p$ab/r_kernel_trace.csv')) if 'k1b' in r['Kernel_Name']]
print('ablate $ab  k1b us: min %.1f median %.1f' % (min(d), sorted(d)[len(d)//2]), {k:v for k,v in agg.items() if 'k1b' in k})
PY
done
