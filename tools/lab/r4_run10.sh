#!/bin/bash
# round 4: hd-32 attention tiles as unpadded swizzled 64-byte rows (A/B build) -- parity, timing, LDS counters
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4; mkdir -p $OUT
SWZ=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_attn32swz.so
MASKDIT_HIP_LIB=$SWZ timeout 900 python -m pytest tests/test_00_kernels_gpu.py -m gpu -q -k "attention" > $OUT/attn32_tests.log 2>&1
echo "attn tests (swz lib) rc=$?" >> $OUT/attn32_tests.log
for i in 1 2; do
  MASKDIT_HIP_LIB=$SWZ timeout 300 python tools/attn_bench.py 2>&1 | grep -E "256, 16, 32\)|shape" > $OUT/attn32_bench_swz_$i.txt
  timeout 300 python tools/attn_bench.py 2>&1 | grep -E "256, 16, 32\)|shape" > $OUT/attn32_bench_prod_$i.txt
done
for tag in swz prod; do
  if [ $tag = swz ]; then export MASKDIT_HIP_LIB=$SWZ; else unset MASKDIT_HIP_LIB; fi
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_attn32_$tag -o p -- python tools/attn_bench.py > $OUT/pmc_attn32_$tag.log 2>&1
  python tools/pmc_table.py $OUT/pmc_attn32_$tag 2>&1 | grep -E "kernel|_kernel<32" | cut -c1-70,155-185 > $OUT/pmc_attn32_$tag.txt
  rm -rf $OUT/pmc_attn32_$tag
done
unset MASKDIT_HIP_LIB
tail -2 $OUT/attn32_tests.log; echo SWZ; cat $OUT/attn32_bench_swz_*.txt; echo PROD; cat $OUT/attn32_bench_prod_*.txt; cat $OUT/pmc_attn32_swz.txt $OUT/pmc_attn32_prod.txt
