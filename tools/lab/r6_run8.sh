#!/bin/bash
# round 6, GPU call 8: software-pipelined fp32 GEMM K loop -- parity + per-shape table + 1-WG/CU probe
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" > $OUT/t8_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/t8_kernels.log
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_c.txt 2>&1; cat $OUT/f32_bench_c.txt | grep -v amdgpu.ids
MDT_F32_LDS_PAD=40960 python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep -v amdgpu
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
