export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], c['matches_total'], c['prefix_hits_per_step'])"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dist U | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('U', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], c['matches_total'], c['prefix_hits_per_step'])"
