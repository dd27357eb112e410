#!/bin/bash
# round 4, call A: the short-pattern side test (parity first), the whole GPU suite, then same-box bench lines
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4a
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu > $OUT/pytest_r4.log 2>&1
echo "round4 tests rc=$?"; tail -15 $OUT/pytest_r4.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round4.py > $OUT/pytest_all.log 2>&1
echo "suite rc=$?"; tail -8 $OUT/pytest_all.log
for cfg in cfg2 mixed mixedx mixedb cfg5 cfg4; do
  timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-target-size --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$cfg.json")); c = d["config"]; r = d["roofline"]
    print("$cfg", d["value"], "GB/s", d["ms_per_step"], "ms/step K1", r["kernel"], r["kernel_ms"], "ms matches", c["matches_total"], "hits", c["prefix_hits_per_step"], "occ", c["raw_occurrences_per_step"], "cold", c["value_no_settle"])
except Exception as e:
    print("$cfg failed", e); print(open("$OUT/bench_$cfg.err").read()[-1500:])
PY
done
ACX_NO_SHORT_SPLIT=1 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-target-size --config mixed > $OUT/bench_mixed_nosplit.json 2> $OUT/bench_mixed_nosplit.err
python -c "
import json; d=json.load(open('$OUT/bench_mixed_nosplit.json')); print('mixed (round-3 path, ACX_NO_SHORT_SPLIT)', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"
