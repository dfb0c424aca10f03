#!/bin/bash
# round 6, GPU call 15: LDS-DMA form of the fp32 GEMM -- parity (kernel + engine fp32 tests) and A/B against the register-staged form
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" > $OUT/t15_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/t15_kernels.log | cut -c1-200
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_dma.txt 2>&1; grep -v amdgpu.ids $OUT/f32_bench_dma.txt | head -9
MDT_F32_DMA=0 timeout 600 python tools/f32_bench.py 2>&1 | grep -v amdgpu.ids | head -9
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
MDT_F32_DMA=0 timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s 2>&1 | grep -i "fp32\|passed\|failed\|error" | tail -8 | cut -c1-260
