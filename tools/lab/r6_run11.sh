#!/bin/bash
# round 6, GPU call 11: ablation of the fp32 GEMM K loop (experiments build; garbage results): what costs the 27 % of idle matrix pipe?
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export MASKDIT_HIP_LIB=$GRAFT_REPO_ROOT/maskdit_amd/libmaskdit_hip_exp.so
for tile in 128 0; do
for ab in 0 1 2 3 4 5 7; do
  echo -n "tile=$tile ablate=$ab: "; MDT_F32_TILE=$tile MDT_F32_ABLATE=$ab python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep TF
done; done
echo "one workgroup per CU:"
for ab in 0 1 7; do
  echo -n "tile=128 pad ablate=$ab: "; MDT_F32_LDS_PAD=40960 MDT_F32_TILE=128 MDT_F32_ABLATE=$ab python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep TF
done
