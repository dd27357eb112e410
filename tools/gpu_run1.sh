cd /root/repo
export TMPDIR=/tmp
TAG=_k1a bash tools/gpu_pmc_ab.sh "--kernel dfa_walk" tree 2>&1 | grep -E "k1a_scan"
