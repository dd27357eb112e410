cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-target-size > gpurun_out/r03/bench_T_one_rank_rccl.json 2> gpurun_out/r03/bench_T_one_rank_rccl.err
grep "^{" gpurun_out/r03/bench_T_one_rank_rccl.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['config']['value_no_settle'],d['config']['collective'])" || tail -5 gpurun_out/r03/bench_T_one_rank_rccl.err
