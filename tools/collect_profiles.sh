#!/bin/bash
# Collect the round's evidence on a GPU box (run from the repo root through gpurun):
#   bench lines (cfg2 T/U with the CPU baseline, the cold number and the 8 GiB target size, cfg3 shape, cfg4,
#   cfg4b, cfg5, host-memory entry point, K1a, the dense path), rocprofv3 kernel-trace stats, PMC passes
#   (HBM traffic, SQ, LDS).  PMC passes are separate runs (FETCH_SIZE costs 3 of the 4 TCC slots) and never
#   combined with tracing other than --kernel-trace.  Every command is bounded by `timeout`.
# usage: tools/collect_profiles.sh [round] [part]   then: python tools/summarize_profiles.py [round]
# (round 5: + the hot pipeline's inputs H1 / H100 / the density sweep, K0's prefilter against the walks, issue-side counters of
#  every K1b variant on file: plain, BIG (cfg4), anchors + code points (cfg5), side test (mixed), STAGED (large))
#   part: all (default) | bench | trace | pmc
# usage: tools/collect_profiles.sh --check [round]   (no GPU) fails when profiles/<round> was measured on other
#        device code than the working tree's (tools/kernel_hash.py), i.e. when the kernels changed since
set -u
export TMPDIR=/tmp
if [ "${1:-}" = "--check" ]; then
  R=${2:-r06}
  cd "$(dirname "$0")/.."
  HAVE=$(cat profiles/$R/kernel_source_sha256.txt 2>/dev/null || echo none)
  NOW=$(python tools/kernel_hash.py)
  if [ "$HAVE" != "$NOW" ]; then
    echo "profiles/$R was measured on device code $HAVE, the working tree is $NOW: collect again"; exit 1
  fi
  echo "profiles/$R matches the working tree's device code ($NOW)"; exit 0
fi
R=${1:-r06}
PART=${2:-all}
OUT=/root/repo/gpurun_out/$R
mkdir -p $OUT
cd /root/repo
python tools/kernel_hash.py > $OUT/kernel_source_sha256.txt
if [ "$PART" = all ] || [ "$PART" = bench ]; then
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_T.json 2> $OUT/bench_T.err
timeout 300 python bench.py --dist U --no-cpu-baseline --no-secondary > $OUT/bench_U.json 2> $OUT/bench_U.err
timeout 300 python bench.py --config cfg3 --no-cpu-baseline > $OUT/bench_cfg3_shape_n1.json 2> $OUT/bench_cfg3.err
for cfg in cfg4 cfg4b cfg5 cfg2b; do
  timeout 900 python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
done
timeout 900 python bench.py --config large --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_large_1M_patterns.json 2> $OUT/bench_large.err
timeout 600 python bench.py --host --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_host_path.json 2> $OUT/bench_host.err
for d in T U; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --dist $d --kernel dfa_walk > $OUT/bench_${d}_dfa_walk.json 2> $OUT/bench_${d}_dfa_walk.err
  ACX_NO_PFAC=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target-size --no-cold --dist $d --kernel dfa_walk > $OUT/bench_${d}_dfa_walk_chunked.json 2> $OUT/bench_${d}_dfa_walk_chunked.err
done
# mixed-length sets (round 4: K1b with its side test): rare short patterns, the round-2 VERDICT's example (dense
# output), cfg5's set + short patterns (> 32 byte classes: the round-3 cliff); and the round-3 path for comparison
for cfg in mixed mixedx mixedb; do
  timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
done
ACX_NO_SHORT_SPLIT=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --config mixed > $OUT/bench_mixed_round3_path_k1a.json 2> $OUT/bench_mixed_round3_path_k1a.err
# anchors off: cfg5 on the round-3 tables
ACX_NO_ANCHORS=1 timeout 600 python bench.py --config cfg5 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg5_no_anchors.json 2> $OUT/bench_cfg5_no_anchors.err
# dense stretches: ONE 64 KiB region / 1 % of the groups (the hot pipeline), and the curve from sparse to dense
for d in H1 H100 P1024 P512 P256 P128 P64 P32; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-target-size --dist $d > $OUT/bench_hot_$d.json 2> $OUT/bench_hot_$d.err
done
# the dense paths: a dense input through the tile-ordered form and through the radix-sort form, the headline input forced onto it
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-target-size --dist D > $OUT/bench_dense_D.json 2> $OUT/bench_dense_D.err
ACX_NO_DENSE_TILES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-target-size --dist D > $OUT/bench_dense_D_radix_form.json 2> $OUT/bench_dense_D_radix_form.err
ACX_NO_BUCKET=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-secondary > $OUT/bench_T_forced_dense_path.json 2> $OUT/bench_T_forced_dense_path.err
# one rank under the launcher: RCCL init + the count all-gather on the device
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-target-size > $OUT/bench_T_one_rank_rccl.json 2> $OUT/bench_T_one_rank_rccl.err
timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison.txt 2>&1
ACX_SMALL_SYNC=1 timeout 600 python benchmarks/bench_comparison.py > $OUT/bench_comparison_stream_sync.txt 2>&1
# round 6: without the resident K0 (a launch per small call) and without the in-place mid-size path, same box
ACX_NO_RESIDENT=1 ACX_INPLACE_MAX=0 timeout 600 python benchmarks/bench_comparison.py < /dev/null > $OUT/exp_no_resident_no_inplace_bench_comparison.txt 2>&1
# K0 without its prefilter mode (round 4's K0: the walks, 16 KiB at most), same box
ACX_K0_NO_PREFILTER=1 timeout 300 python benchmarks/bench_comparison.py < /dev/null > $OUT/exp_k0_no_prefilter_bench_comparison.txt 2>&1
for ds in short short_nomatch short_onematch long; do timeout 60 python tools/k0_probe.py $ds indexes 2000 < /dev/null; done > $OUT/k0_probe.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
[ -x tools/ubench_stream.bin ] && timeout 120 tools/ubench_stream.bin > $OUT/ubench_stream.txt 2>&1
fi
if [ "$PART" = all ] || [ "$PART" = trace ]; then
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --no-secondary > $OUT/trace_T.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T_8gib -o bench -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-target-size --no-cold --no-secondary --bytes 8589934592 > $OUT/trace_T_8gib.log 2>&1
for cfg in cfg4 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --config $cfg > $OUT/trace_$cfg.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dfa_walk -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-target-size --no-cold --kernel dfa_walk > $OUT/trace_dfa_walk.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dense_D -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist D > $OUT/trace_dense_D.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mixed -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-target-size --config mixed > $OUT/trace_mixed.log 2>&1
for d in H1 H100; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$d -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-secondary --no-target-size --dist $d > $OUT/trace_$d.log 2>&1
done
# the resident K0 under the tracer: the reference's short loop (its kernels' lives are what the trace shows)
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k0_short -o bench -- python /root/repo/tools/k0_probe.py short indexes 5000 > $OUT/trace_k0_short.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_large -o bench -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cold --config large > $OUT/trace_large.log 2>&1
cd /root/repo
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
cd /tmp
P="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-target-size --no-cold --no-secondary"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- $P > $OUT/pmc_write.log 2>&1
# calibration of FETCH_SIZE on this access pattern: the same kernels over a haystack of zero bytes
# read exactly 1 GiB and do nothing else (no survivor, no gather, no hit)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_zero_haystack -o r -- $P --dist Z > $OUT/pmc_fetch_zero.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -o r -- $P > $OUT/pmc_tcc.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o r -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o r -- $P > $OUT/pmc_lds.log 2>&1
# K1a (the failureless walk)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_dfa_walk -o r -- $P --kernel dfa_walk > $OUT/pmc_fetch_dfa_walk.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_dfa_walk -o r -- $P --kernel dfa_walk > $OUT/pmc_write_dfa_walk.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_dfa_walk_zero_haystack -o r -- $P --kernel dfa_walk --dist Z > $OUT/pmc_fetch_dfa_walk_zero.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq_dfa_walk -o r -- $P --kernel dfa_walk > $OUT/pmc_sq_dfa_walk.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds_dfa_walk -o r -- $P --kernel dfa_walk > $OUT/pmc_lds_dfa_walk.log 2>&1
# the other BASELINE configurations: HBM traffic of their scan kernel, and its issue-side counters (round 5: every K1b
# variant on file -- BIG, anchors + code points, the side test, STAGED)
for cfg in cfg4 cfg5; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$cfg -o r -- $P --config $cfg > $OUT/pmc_fetch_$cfg.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$cfg -o r -- $P --config $cfg > $OUT/pmc_write_$cfg.log 2>&1
  timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc_$cfg -o r -- $P --config $cfg > $OUT/pmc_tcc_$cfg.log 2>&1
done
for cfg in cfg4 cfg5 mixed large; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq_$cfg -o r -- $P --config $cfg > $OUT/pmc_sq_$cfg.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds_$cfg -o r -- $P --config $cfg > $OUT/pmc_lds_$cfg.log 2>&1
done
cd /root/repo
fi
ls $OUT | head -80
cut -c1-400 $OUT/bench_T.json 2>/dev/null
