cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_entry_gpu.py -x -q -m gpu -k "two_ranks_zero1" 2>&1 | grep -v "^$" | tail -30
