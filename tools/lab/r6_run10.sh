#!/bin/bash
# round 6, GPU call 10: fp32 GEMM 256 x 128 x 16 tile vs 128 x 128 x 32 (MDT_F32_TILE=128)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" > $OUT/t10_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/t10_kernels.log
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_e.txt 2>&1; cat $OUT/f32_bench_e.txt | grep -v amdgpu.ids
MDT_F32_TILE=128 timeout 600 python tools/f32_bench.py 2>&1 | grep -v amdgpu.ids | head -9
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s 2>&1 | grep -i "fp32\|passed\|failed\|error" | tail -8 | cut -c1-260
