#!/bin/bash
# Collect the round's evidence on a GPU box (run from the repo root through gpurun):
#   bench lines (cfg2 T/U with the CPU baseline, cfg3 shape, cfg4, cfg4b, cfg5, 8 GiB, host-memory
#   entry point, K1a), rocprofv3 kernel-trace stats, PMC passes (HBM traffic, SQ, LDS).
# PMC passes are separate runs (FETCH_SIZE costs 3 of the 4 TCC slots) and never combined
# with tracing other than --kernel-trace.  Every command is bounded by `timeout`.
# usage: tools/collect_profiles.sh [round]     then: python tools/summarize_profiles.py [round]
set -u
export TMPDIR=/tmp
R=${1:-r02}
OUT=/root/repo/gpurun_out/$R
rm -rf $OUT; mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py > $OUT/bench_T.json 2> $OUT/bench_T.err
timeout 300 python bench.py --dist U --no-cpu-baseline > $OUT/bench_U.json 2> $OUT/bench_U.err
timeout 300 python bench.py --config cfg3 --no-cpu-baseline > $OUT/bench_cfg3_shape_n1.json 2> $OUT/bench_cfg3.err
for cfg in cfg4 cfg4b cfg5; do
  timeout 900 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
done
timeout 900 python bench.py --config large --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_large_1M_patterns.json 2> $OUT/bench_large.err
timeout 900 python bench.py --bytes 8589934592 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_T_8GiB.json 2> $OUT/bench_8g.err
timeout 600 python bench.py --host --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_host_path.json 2> $OUT/bench_host.err
for m in 2 3; do ACX_STAGE=$m timeout 600 python bench.py --host --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_host_path_stage$m.json 2>> $OUT/bench_host.err; done
for d in T U; do
  ACX_WALK_STATS=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --dist $d --kernel dfa_walk > $OUT/bench_${d}_dfa_walk.json 2> $OUT/bench_${d}_dfa_walk.err
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_T.log 2>&1
for cfg in cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --config $cfg > $OUT/trace_$cfg.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dfa_walk -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --kernel dfa_walk > $OUT/trace_dfa_walk.log 2>&1
P="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- $P > $OUT/pmc_write.log 2>&1
# calibration of FETCH_SIZE on this access pattern: the same kernel over a haystack of zero bytes
# reads exactly 1 GiB and does nothing else (no survivor, no gather, no hit)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_zero_haystack -o r -- $P --dist Z > $OUT/pmc_fetch_zero.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o r -- $P > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o r -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o r -- $P > $OUT/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_dfa_walk -o r -- $P --kernel dfa_walk > $OUT/pmc_fetch_dfa_walk.log 2>&1
cd /root/repo
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
cut -c1-300 $OUT/bench_T.json
