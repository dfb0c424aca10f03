#!/bin/bash
# round 6, GPU call 2: fp32 GEMM per-shape table, fp32 sampler kernel trace, tn8 fixed-cost fit
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_a.txt 2>&1; cat $OUT/f32_bench_a.txt | grep -v amdgpu.ids
timeout 900 python tools/tn8_fixed_cost.py > $OUT/tn8_fixed_cost.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_fixed_cost.txt | grep "fit\|M =\|auto\|---"
rocprofv3 --kernel-trace --stats -d $OUT/kt_f32 -o kt -- python tools/sampler_profile.py 3 fp32 > $OUT/kt_f32.log 2>&1
db=$(find $OUT/kt_f32 -name "*.db" | head -1)
python tools/rocprof_summary.py $db $OUT/kernel_stats_sampler_fp32_a.txt "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 3 fp32 (first build of the fp32 path)" > /dev/null
rm -rf $OUT/kt_f32
head -24 $OUT/kernel_stats_sampler_fp32_a.txt | cut -c1-170
