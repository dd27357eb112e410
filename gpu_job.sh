export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q 2>&1 | tail -4
for ab in 0 2 1 5; do
 echo "ABLATE=$ab"
 ACX_ABLATE=$ab python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('T', d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['config']['matches_total'])"
 ACX_ABLATE=$ab python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dist U | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('U', d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['config']['matches_total'])"
done
