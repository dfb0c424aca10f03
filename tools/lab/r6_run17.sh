#!/bin/bash
# round 6, GPU call 17: LDS-DMA fp32 GEMM with the DMA issued at the top of the K-tile; parity + table + sampler + A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" 2>&1 | tail -1
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_dma2.txt 2>&1; grep -v amdgpu.ids $OUT/f32_bench_dma2.txt
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
MDT_F32_DMA=0 timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s 2>&1 | grep -i "fp32\|passed\|failed\|error" | tail -8 | cut -c1-260
