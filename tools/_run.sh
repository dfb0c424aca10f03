cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -2
python tools/nt8_bench.py --iters 3 --rounds 2 --decoder 2>&1 | cut -c1-75
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > gpurun_out/r3/bench_f.json 2> gpurun_out/r3/bench_f.err
cut -c1-330 gpurun_out/r3/bench_f.json; grep -o '"roofline.*' gpurun_out/r3/bench_f.json | cut -c1-300
