cd /root/repo
export TMPDIR=/tmp
TAG=_g bash tools/gpu_trace_ab.sh "" z g32t256 g32t192 g32t128 g48t256 z 2>&1 | grep -E "==|k1b_prefilter<|k_tile_main   |k_tile_write   "
for v in z g32t256 g32t192 g48t256; do ACX_LIB=/root/repo/variants/libacx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-target-size --no-cold | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['ms_per_step'],d['config']['matches_total'])"; done
