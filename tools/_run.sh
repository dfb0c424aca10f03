cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -4
python tools/nt8_sched.py > gpurun_out/r3/sched7.log 2>&1
cat gpurun_out/r3/sched7.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > gpurun_out/r3/bench_d.json 2> gpurun_out/r3/bench_d.err
cut -c1-400 gpurun_out/r3/bench_d.json; grep -o '"roofline.*' gpurun_out/r3/bench_d.json | cut -c1-300
