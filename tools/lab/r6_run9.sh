#!/bin/bash
# round 6, GPU call 9: grouped tile order for the fp32 GEMM -- parity + per-shape table + PMC (L2 hit rate, MFMA busy)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32" > $OUT/t9_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/t9_kernels.log
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_d.txt 2>&1; cat $OUT/f32_bench_d.txt | grep -v amdgpu.ids
MDT_F32_LDS_PAD=40960 python tools/f32_one.py 4608 1152 NONE 3 2>&1 | grep -v amdgpu
CMD="python tools/f32_one.py 4608 1152 NONE 3"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pf_write -o p -- $CMD > $OUT/pf_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pf_sq -o p -- $CMD > $OUT/pf_sq.log 2>&1
python tools/pmc_table.py $OUT/pf_write $OUT/pf_sq > $OUT/f32_pmc_b.txt 2>&1
rm -rf $OUT/pf_write $OUT/pf_sq
grep "gemm_f32\|kernel " $OUT/f32_pmc_b.txt | cut -c1-260
timeout 600 python tools/sampler_profile.py 10 fp32 2>&1 | grep -v amdgpu.ids
