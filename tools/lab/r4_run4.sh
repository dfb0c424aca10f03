#!/bin/bash
# round 4, fourth GPU call: the conflict-free hd-72 attention image -- parity tests, same-box A/B timing against the
# 144-byte-row build, LDS counters of both
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k "attention" > $OUT/attn_tests.log 2>&1
echo "attn tests rc=$?" >> $OUT/attn_tests.log
timeout 600 python -m pytest tests/test_10_engine_gpu.py -m gpu -q -k "baseline_configs or s2_train_step or forward_loss" > $OUT/attn_e2e.log 2>&1
echo "e2e rc=$?" >> $OUT/attn_e2e.log
timeout 300 python tools/attn_bench.py > $OUT/attn_bench_split.txt 2>&1
MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_attn144.so timeout 300 python tools/attn_bench.py > $OUT/attn_bench_144.txt 2>&1
for tag in split 144; do
  if [ $tag = 144 ]; then export MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_attn144.so; else unset MASKDIT_HIP_LIB; fi
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_attn_$tag -o p -- python tools/attn_bench.py > $OUT/pmc_attn_$tag.log 2>&1
  python tools/pmc_table.py $OUT/pmc_attn_$tag 2>&1 | grep -E "kernel|attn" | cut -c1-260 > $OUT/pmc_attn_$tag.txt
  rm -rf $OUT/pmc_attn_$tag
done
unset MASKDIT_HIP_LIB
tail -3 $OUT/attn_tests.log; tail -3 $OUT/attn_e2e.log; echo SPLIT; cat $OUT/attn_bench_split.txt; echo OLD144; cat $OUT/attn_bench_144.txt; cat $OUT/pmc_attn_split.txt; cat $OUT/pmc_attn_144.txt
