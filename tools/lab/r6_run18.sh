#!/bin/bash
# round 6, GPU call 18: per-shape table + PMC of the final fp32 GEMM (LDS-DMA form, DMA at the top of the K-tile) vs the register-staged form
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 600 python tools/f32_bench.py > $OUT/f32_bench_dma2.txt 2>&1; grep -v amdgpu.ids $OUT/f32_bench_dma2.txt
MDT_F32_DMA=0 timeout 600 python tools/f32_bench.py 2>&1 | grep -v amdgpu.ids | sed -n 2,9p > $OUT/f32_bench_reg2.txt; cat $OUT/f32_bench_reg2.txt
CMD="python tools/f32_one.py 4608 1152 NONE 3"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pf_write -o p -- $CMD > $OUT/pf_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pf_sq -o p -- $CMD > $OUT/pf_sq.log 2>&1
python tools/pmc_table.py $OUT/pf_write $OUT/pf_sq > $OUT/f32_pmc_c.txt 2>&1
rm -rf $OUT/pf_write $OUT/pf_sq
grep "gemm_f32\|kernel " $OUT/f32_pmc_c.txt | cut -c1-260
