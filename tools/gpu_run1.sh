cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py tests/test_gpu_cfg1.py tests/test_gpu_compressed.py -x -q -m gpu 2>&1 | tail -15
for d in T U; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --dist $d --kernel dfa_walk > gpurun_out/k1a_$d.json 2> gpurun_out/k1a_$d.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/k1a_$d.json")); r = d["roofline"]
    print("K1a $d:", d["ms_per_step"], "ms/step", d["value"], "GB/s  scan", r["kernel_ms"], "ms  matches", d["config"]["matches_total"])
except Exception as e:
    print("failed", e); print(open("gpurun_out/k1a_$d.err").read()[-1500:])
PY
done
TAG=_k1a bash tools/gpu_trace_ab.sh "--kernel dfa_walk" tree 2>&1 | grep -E "k1a|k_tile|k_walk" | head -6
TAG=_k1a bash tools/gpu_pmc_ab.sh "--kernel dfa_walk" tree 2>&1 | grep -E "k1a_scan"
