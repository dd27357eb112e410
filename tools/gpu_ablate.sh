#!/bin/bash
# walk-kernel experiments (timing only; ablated results are wrong by construction)
set -u
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/ablate
mkdir -p $OUT
cd /tmp
ACX_ABLATE=128 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ab128 -o bench -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ab128.log 2>&1
echo "== ablate 128"; python /root/repo/tools/rocprof_summary.py $OUT/ab128 2>/dev/null | grep -E "walk|k1b|tile"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o r -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1
ls $OUT/pmc_sq $OUT/pmc_sq2
