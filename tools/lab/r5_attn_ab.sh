#!/bin/bash
# same-box A/B of the attention kernels: round-4 attention.hip (libmaskdit_hip_r4attn.so) vs the round-5 library
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
{
echo "# tools/attn_bench.py, SAME BOX, same process order: round-4 attention.hip (commit c8f5dfc, linked with the round-5 objects) then the round-5 library, twice"
for rep in 1 2; do
echo "## round-4 attention kernels (pass $rep)"
MASKDIT_HIP_LIB=maskdit_amd/libmaskdit_hip_r4attn.so python tools/attn_bench.py 2>&1 | grep -v amdgpu
echo "## round-5 attention kernels (pass $rep)"
python tools/attn_bench.py 2>&1 | grep -v amdgpu
done
} > $OUT/attn_ab_same_box.txt 2>&1
cat $OUT/attn_ab_same_box.txt
