cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py tests/test_gpu_cfg1.py tests/test_gpu_compressed.py tests/test_gpu_round3.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -8
for c in mixed mixedx; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --config $c > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err
  python -c "import json;d=json.load(open('gpurun_out/bench_$c.json'));print('$c',d['value'],d['ms_per_step'],d['config']['scan_kernel'],d['config']['matches_total'],d['roofline']['kernel'],d['roofline']['kernel_ms'])" || tail -5 gpurun_out/bench_$c.err
done
TAG=_mx bash tools/gpu_trace_ab.sh "--config mixedx" tree 2>&1 | head -24
