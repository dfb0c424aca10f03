cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > gpurun_out/r3/bench_g.json 2> gpurun_out/r3/bench_g.err
cut -c1-330 gpurun_out/r3/bench_g.json; grep -o '"roofline.*' gpurun_out/r3/bench_g.json | cut -c1-300
