cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -5 > gpurun_out/r3/t_gemm.log
cat gpurun_out/r3/t_gemm.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-sampler > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err
cat gpurun_out/r3/bench_b.json | cut -c1-1500
