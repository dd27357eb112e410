#!/bin/bash
# kernel trace of tools/measure_configs.py
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/misc
rm -rf $OUT/cfgtrace; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfgtrace -o t -- python /root/repo/tools/measure_configs.py 256 > $OUT/cfgtrace.log 2>&1
tail -5 $OUT/cfgtrace.log
python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/cfgtrace/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# print the last pass of each automaton: find k1b/k1a launches and following kernels
names=[(r['Kernel_Name'].split('(')[0].replace('acx::','').replace('void ','')[:32], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
# group by sequences starting at a scan kernel
seqs=[]; cur=None
for n,d in names:
    if n.startswith('k1b_prefilter') or n.startswith('k1a_dfa_walk'):
        if cur: seqs.append(cur)
        cur=[(n,d)]
    elif cur is not None:
        cur.append((n,d))
if cur: seqs.append(cur)
for idx in (6, 13, len(seqs)-1):
    if idx < len(seqs):
        print('--- pass', idx, ' '.join(f'{n}:{d:.0f}' for n,d in seqs[idx][:14]))
PY
