#!/bin/bash
# round 4, call T: the bench lines that carry roofline.traffic (PMC file of this device code now in the tree), then
# randomised parity on the final kernels (two new seeds)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/r04
mkdir -p $OUT
cd /root/repo
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null > $OUT/bench_T.json 2> $OUT/bench_T.err
echo "bench T rc=$?"; cut -c1-120 $OUT/bench_T.json < /dev/null
for cfg in cfg4 cfg5; do
  timeout 200 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline < /dev/null > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  echo "bench $cfg rc=$?"
done
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --dist T --kernel dfa_walk < /dev/null > $OUT/bench_T_dfa_walk.json 2> $OUT/bench_T_dfa_walk.err
for seed in 40404 40405; do
  timeout 200 python tools/gpu_fuzz.py 100 $seed < /dev/null > $OUT/fuzz_$seed.log 2>&1
  echo "fuzz seed $seed rc=$?"; tail -2 $OUT/fuzz_$seed.log < /dev/null
done
