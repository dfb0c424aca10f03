#!/bin/bash
# round 6, GPU call 16: LDS-DMA fp32 GEMM tile variants (MDT_F32_DMA = 1: 128x128x32, 2: 256x128x16, 3: 128x128x16 with up to 4 workgroups per CU)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
for v in 2 3; do
MDT_F32_DMA=$v timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "gemm_f32" 2>&1 | tail -1
done
for v in 1 2 3; do
echo "== MDT_F32_DMA=$v"; MDT_F32_DMA=$v timeout 600 python tools/f32_bench.py 2>&1 | grep -v amdgpu.ids | sed -n 2,9p
MDT_F32_DMA=$v timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
done
