#!/bin/bash
# construction at scale: the large pattern-set config, and the compressed form beside the dense one
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/f3
mkdir -p $OUT
cd /root/repo
run() { tag=$1; shift; timeout 600 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; python - <<PY
import json
try:
    d = json.load(open("$OUT/$tag.json")); c = d["config"]
    print("$tag", d["value"], "GB/s", d["ms_per_step"], "ms/step build_s", c.get("build_s"), "table", c.get("dfa_table_bytes"), "kernel", c.get("scan_kernel"), "matches", c["matches_total"], "path", c.get("output_path"))
except Exception as e:
    print("$tag failed", e); print(open("$OUT/$tag.err").read()[-800:])
PY
}
run large python bench.py --config large --steps 10 --warmup 2 --no-cpu-baseline
run cfg4b_compressed python bench.py --config cfg4b --steps 10 --warmup 2 --no-cpu-baseline
ACX_DENSE_LIMIT=4294967296 run cfg4b_dense python bench.py --config cfg4b --steps 10 --warmup 2 --no-cpu-baseline
run cfg2_walk_dense python bench.py --kernel dfa_walk --steps 5 --warmup 1 --no-cpu-baseline
ACX_DENSE_LIMIT=0 run cfg2_walk_compressed python bench.py --kernel dfa_walk --steps 5 --warmup 1 --no-cpu-baseline
ACX_DENSE_LIMIT=0 run cfg2_T_compressed python bench.py --steps 10 --warmup 2 --no-cpu-baseline
