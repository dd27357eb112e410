for ab in 0 16 0 16; do
 ACX_ABLATE=$ab python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('T ab=$ab', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved'])"
done
