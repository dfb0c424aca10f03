#!/bin/bash
# round 6, GPU call 14: the residual-in-LayerNorm form in INFERENCE plans (bf16 sampler) A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do
python tools/sampler_profile.py 20 2>&1 | grep -v amdgpu
MDT_FUSE_RES_LN_EVAL=1 python tools/sampler_profile.py 20 2>&1 | grep -v amdgpu
done
MDT_FUSE_RES_LN_EVAL=1 timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "sampler or eval" 2>&1 | tail -2
