#!/bin/bash
# soak: the kernel-level GPU tests three times in fresh processes + the full-batch tests once (flakiness check before the driver's run)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6soak
mkdir -p $OUT
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q > $OUT/kernels_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $OUT/kernels_$i.log)"
done
timeout 900 python -m pytest tests/test_40_full_batch_gpu.py tests/test_10_engine_gpu.py -x -q > $OUT/full_1.log 2>&1; echo "full rc=$? $(tail -1 $OUT/full_1.log)"
