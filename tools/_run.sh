cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
MDT_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29877 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-sampler --global-batch 256 > gpurun_out/r3/bench2.log 2>&1
grep -v "^$" gpurun_out/r3/bench2.log | tail -40
