#!/bin/bash
# round 4, call W: the non-overlapping view of automata with copies of a pattern (host tables only) -- the tests that
# reach it: the new one, cfg1 (names with ~5 % duplicates), the short-pattern sets with duplicates, a slice of the fuzz
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r4w
mkdir -p $OUT
cd /root/repo
timeout 100 python -m pytest tests/test_gpu_round4.py tests/test_gpu_cfg1.py -x -q -k "copies or short_patterns or cfg1 or prose or k0" < /dev/null > $OUT/pytest.log 2>&1
echo "tests rc=$?"; tail -4 $OUT/pytest.log < /dev/null
