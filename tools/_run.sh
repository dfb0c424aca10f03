cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py tests/test_ddp_gpu.py -x -q -m gpu 2>&1 | tail -3
