export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], c['matches_total'], c['prefix_hits_per_step'])"
timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dist U | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('U', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], c['matches_total'], c['prefix_hits_per_step'])"
cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/trace_r1 -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/trace_r1.log 2>&1
