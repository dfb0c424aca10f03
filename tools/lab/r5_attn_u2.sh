#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
{
for rep in 1 2; do
echo "## product (unroll 1) pass $rep"; python tools/attn_bench.py 2>&1 | grep "512\|1024, 16"
echo "## unroll 2 pass $rep"; MASKDIT_HIP_LIB=maskdit_amd/libmaskdit_hip_u2.so python tools/attn_bench.py 2>&1 | grep "512\|1024, 16"
done
} > $OUT/attn_u2.txt 2>&1
cat $OUT/attn_u2.txt
