cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err
cut -c1-1500 gpurun_out/r3/bench_c.json
