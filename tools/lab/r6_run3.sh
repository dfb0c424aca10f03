#!/bin/bash
# round 6, GPU call 3: fused fp32 attention + register softmax (parity, per-shape table, sampler trace), tn8 fixed-cost fit
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" -s > $OUT/t3_kernels.log 2>&1; echo "kernels rc=$?"
grep "attention_f32\|passed\|failed\|rror" $OUT/t3_kernels.log | cut -c1-200 | tail -14
timeout 1200 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s > $OUT/t3_engine.log 2>&1; echo "engine rc=$?"
grep -i "fp32\|passed\|failed\|error" $OUT/t3_engine.log | tail -8 | cut -c1-260
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_b.txt 2>&1; cat $OUT/f32_bench_b.txt | grep -v amdgpu.ids
timeout 900 python tools/tn8_fixed_cost.py > $OUT/tn8_fixed_cost.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_fixed_cost.txt
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
