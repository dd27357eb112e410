#!/bin/bash
# A/B of library variants on ONE box (box-to-box spread is +-3 %): variants/libacx_*.so against the in-tree build.
# usage: tools/gpu_ab.sh "<bench args>" name1 name2 ...   ("tree" = the in-tree library)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ab
mkdir -p $OUT
cd /root/repo
ARGS=${1:-}
shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset ACX_LIB; else export ACX_LIB=/root/repo/variants/libacx_$v.so; export ACX_LIB_ANY_ABI=1; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold $ARGS > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/${v}_$rep.json")); r = d["roofline"]
    print("$v rep $rep:", d["ms_per_step"], "ms/step", d["value"], "GB/s  K1", r["kernel_ms"], "ms  matches", d["config"]["matches_total"])
except Exception as e:
    print("$v failed", e); print(open("$OUT/${v}_$rep.err").read()[-800:])
PY
  done
done
unset ACX_LIB
