for i in 1 2; do
timeout 100 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('T', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'], c['matches_total'])"
done
