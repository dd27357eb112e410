cd /root/repo
export TMPDIR=/tmp
TAG=_sb bash tools/gpu_trace_ab.sh "" dpfac sb dpfac sb 2>&1 | grep -E "==|k1b_prefilter<|k_tile_main   "
TAG=_sbk bash tools/gpu_trace_ab.sh "--kernel dfa_walk" dpfac sb 2>&1 | grep -E "==|k1a_scan|k1a_walk  "
