#!/bin/bash
# PMC counters of the scan kernel for library variants.  usage: tools/gpu_pmc_ab.sh "<bench args>" name...
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_ab${TAG:-}
mkdir -p $OUT
ARGS=${1:-}
shift
SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
      "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE")
for v in "$@"; do
  if [ "$v" = tree ]; then unset ACX_LIB; else export ACX_LIB=/root/repo/variants/libacx_$v.so; fi
  i=0
  for set in "${SETS[@]}"; do
    rm -rf $OUT/${v}_$i
    cd /tmp
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${v}_$i -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-size --no-cold --settle-ms 0 $ARGS > $OUT/${v}_$i.log 2>&1
    cd /root/repo
    i=$((i+1))
  done
  python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${v}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("acx::", "")[:28]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, vv in agg.items():
    if k.startswith(("k1b", "k1a", "k_walk")):
        print("$v", k, {c: round(sum(x) / len(x)) for c, x in sorted(vv.items())})
PY
done
