#!/bin/bash
# round 4: inline-asm transpose reads in attn_bwd_dma (no mid-item vmcnt(0)): parity + stress + same-box A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k "attention" > $OUT/attn2_tests.log 2>&1
echo "attn tests rc=$?" >> $OUT/attn2_tests.log
MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_r3wait.so timeout 600 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k first_item_stress > $OUT/attn2_stress_r3wait.log 2>&1
echo "r3-wait rc=$?" >> $OUT/attn2_stress_r3wait.log
timeout 600 python -m pytest tests/test_40_full_batch_gpu.py tests/test_10_engine_gpu.py -m gpu -q -k "linear_in_slices or baseline_configs" > $OUT/attn2_e2e.log 2>&1
echo "e2e rc=$?" >> $OUT/attn2_e2e.log
for i in 1 2; do
timeout 300 python tools/attn_bench.py 2>&1 | grep -E "128, 16, 72\)|shape" | grep -E "default|shape" > $OUT/attn2_bench_new_$i.txt
MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_attnold.so timeout 300 python tools/attn_bench.py 2>&1 | grep -E "128, 16, 72\)|shape" | grep -E "default|shape" > $OUT/attn2_bench_old_$i.txt
done
tail -2 $OUT/attn2_tests.log; tail -2 $OUT/attn2_stress_r3wait.log; tail -2 $OUT/attn2_e2e.log; echo NEW; cat $OUT/attn2_bench_new_*.txt; echo OLD; cat $OUT/attn2_bench_old_*.txt
