cd /root/repo
export TMPDIR=/tmp
TAG=_k1b bash tools/gpu_trace_ab.sh "" pre cmp tree 2>&1 | grep -E "^==|k1b_pref|k_tile_main|k_tile_write  " | grep -v "^ " 
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py tests/test_api_gpu.py -x -q -m gpu 2>&1 | tail -4
