cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_entry_gpu.py tests/test_kernels_gpu.py tests/test_vae_gpu.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3/t_all2.log
cat gpurun_out/r3/t_all2.log
cd /tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler > $GRAFT_REPO_ROOT/gpurun_out/r3/kt.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r3/kt -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r3/kernel_stats_mid.txt "rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler (round 3, mid: fine-interleaved nt8 + tn8; 5 optimizer steps + plan construction)" | cut -c1-180
rm -rf gpurun_out/r3/kt
