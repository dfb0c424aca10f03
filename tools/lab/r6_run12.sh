#!/bin/bash
# round 6, GPU call 12: fp32 GEMM v2 (two-ahead register staging, 16-byte epilogue) parity + table + ablation
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" > $OUT/t12_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $OUT/t12_kernels.log | cut -c1-200
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_f.txt 2>&1; cat $OUT/f32_bench_f.txt | grep -v amdgpu.ids
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s 2>&1 | grep -i "fp32\|passed\|failed\|error" | tail -8 | cut -c1-260
