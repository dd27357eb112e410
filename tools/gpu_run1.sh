cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -5
TAG=_dw bash tools/gpu_trace_ab.sh "" pre d1w16 d2w16 d2w12 d1w12 pre d1w16 2>&1 | grep -E "==|k1b_prefilter<|k_tile_main   "
