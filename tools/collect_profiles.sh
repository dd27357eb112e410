#!/bin/bash
# Collect the round's evidence on a GPU box (run from the repo root through gpurun):
#   bench line (T and U), rocprofv3 kernel-trace stats, PMC passes for HBM traffic.
# PMC passes are separate runs (FETCH_SIZE costs 3 of the 4 TCC slots) and never combined
# with tracing other than --kernel-trace.  Every command is bounded by `timeout`.
set -u
export TMPDIR=/tmp
R=${1:-r01}
OUT=/root/repo/gpurun_out/$R
mkdir -p $OUT
cd /root/repo
timeout 200 python bench.py --steps 20 --warmup 3 > $OUT/bench_T.json 2> $OUT/bench_T.err
timeout 200 python bench.py --steps 20 --warmup 3 --dist U --no-cpu-baseline > $OUT/bench_U.json 2> $OUT/bench_U.err
timeout 200 python bench.py --steps 10 --warmup 3 --kernel dfa_walk --no-cpu-baseline > $OUT/bench_T_dfa_walk.json 2> $OUT/bench_T_dfa_walk.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_T -o bench -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_T.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
ACX_ABLATE=5 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_loads_only -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_loads_only.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_sq -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_lds.log 2>&1
cd /root/repo
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
cat $OUT/bench_T.json | cut -c1-400
