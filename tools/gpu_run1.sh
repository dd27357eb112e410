cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse_path.py tests/test_gpu_batch.py tests/test_gpu_cfg1.py tests/test_gpu_compressed.py tests/test_gpu_round3.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -8
TAG=_k1a bash tools/gpu_trace_ab.sh "--kernel dfa_walk" sym tree 2>&1 | grep -E "==|k1a_scan|k1a_walk  |k_tile_main   |k_tile_write   |fillBuffer"
TAG=_k1aU bash tools/gpu_trace_ab.sh "--kernel dfa_walk --dist U" sym tree 2>&1 | grep -E "==|k1a_scan|k1a_walk  "
