cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_round3.py tests/test_api_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold --config cfg3 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg3',d['value'],d['ms_per_step'])"; done
ACX_LIB=/root/repo/variants/libacx_z.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold --config cfg3 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg3 before',d['value'],d['ms_per_step'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold | python -c "import json,sys;d=json.loads(sys.stdin.read());print('cfg2',d['value'],d['ms_per_step'])"
