#!/bin/bash
# round 4, first GPU call: full -m gpu suite on the fixed build, the first-item stress test on the regression build
# (must FAIL there), the batch-1024 linearity test three times in one process lease, default bench line
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -rs --durations=15 > $OUT/gputests1.log 2>&1
echo "suite rc=$?" >> $OUT/gputests1.log
MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_r3wait.so timeout 600 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k first_item_stress > $OUT/stress_on_r3_wait.log 2>&1
echo "r3-wait rc=$?" >> $OUT/stress_on_r3_wait.log
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_40_full_batch_gpu.py -m gpu -q -k linear_in_slices -s > $OUT/linear_rep$i.log 2>&1
  echo "rep $i rc=$?" >> $OUT/linear_rep$i.log
done
timeout 900 python bench.py > $OUT/bench1.json 2> $OUT/bench1.err
tail -3 $OUT/gputests1.log; tail -3 $OUT/stress_on_r3_wait.log; tail -2 $OUT/linear_rep*.log; cut -c1-400 $OUT/bench1.json
