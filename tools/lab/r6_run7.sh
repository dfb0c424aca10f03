#!/bin/bash
# round 6, GPU call 7: how many gemm_f32 workgroups does a CU really hold?  (occupancy API + forced LDS padding)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MDT_F32_DEBUG=1 python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep -v amdgpu
MDT_F32_DEBUG=1 MDT_F32_LDS_PAD=40960 python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep -v amdgpu
MDT_F32_DEBUG=1 MDT_F32_LDS_PAD=16384 python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep -v amdgpu
