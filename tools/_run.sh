cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
