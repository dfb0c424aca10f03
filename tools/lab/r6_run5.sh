#!/bin/bash
# round 6, GPU call 5: fp32 MFMA practical roof, fp32 plan at T = 1024, sampler trace (fp32), tn8 model re-check
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f32_peak tools/micro/mfma_f32_peak.hip && /tmp/mfma_f32_peak > $OUT/mfma_f32_peak.txt 2>&1; cat $OUT/mfma_f32_peak.txt
timeout 900 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32" -s > $OUT/t5_engine.log 2>&1; echo "engine rc=$?"
grep -i "fp32\|passed\|failed\|error" $OUT/t5_engine.log | tail -8 | cut -c1-260
timeout 600 python tools/tn8_ab.py 131072 0,256 > $OUT/tn8_ab2_131072.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_ab2_131072.txt | cut -c1-200
timeout 600 python tools/tn8_ab.py 16384 0,256 > $OUT/tn8_ab2_16384.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_ab2_16384.txt | cut -c1-200
rocprofv3 --kernel-trace --stats -d $OUT/kt_f32 -o kt -- python tools/sampler_profile.py 3 fp32 > $OUT/kt_f32.log 2>&1
db=$(find $OUT/kt_f32 -name "*.db" | head -1)
python tools/rocprof_summary.py $db $OUT/kernel_stats_sampler_fp32_b.txt "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 3 fp32 (fused fp32 attention)" > /dev/null
rm -rf $OUT/kt_f32
head -14 $OUT/kernel_stats_sampler_fp32_b.txt | cut -c1-170
