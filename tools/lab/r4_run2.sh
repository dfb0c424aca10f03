#!/bin/bash
# round 4, second GPU call: the strengthened first-item stress test on the product library (must pass) and on the
# regression build with the round-3 wait (must FAIL); the batch-1024 linearity test 10x in one lease
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -s -k first_item_stress > $OUT/stress_product.log 2>&1
echo "product rc=$?" >> $OUT/stress_product.log
MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_r3wait.so timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k first_item_stress > $OUT/stress_on_r3_wait.log 2>&1
echo "r3-wait rc=$?" >> $OUT/stress_on_r3_wait.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python -m pytest tests/test_40_full_batch_gpu.py -m gpu -q -k linear_in_slices -s 2>&1 | grep -E "worst|passed|failed|differ" >> $OUT/linear_10x.log
done
grep -E "passed|failed|rc=" $OUT/stress_product.log $OUT/stress_on_r3_wait.log; cat $OUT/linear_10x.log
