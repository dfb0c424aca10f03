#!/bin/bash
# round 6, GPU call 6: PMC passes on the fp32 GEMM (fc1 shape)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
CMD="python tools/f32_one.py 4608 1152 NONE 3"
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pf_write -o p -- $CMD > $OUT/pf_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/pf_sq -o p -- $CMD > $OUT/pf_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pf_lds -o p -- $CMD > $OUT/pf_lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pf_fetch -o p -- $CMD > $OUT/pf_fetch.log 2>&1
python tools/pmc_table.py $OUT/pf_write $OUT/pf_sq $OUT/pf_lds > $OUT/f32_pmc_a.txt 2>&1
python tools/pmc_dump.py $OUT/pf_fetch 2>&1 | head -5
python - <<'PY'
import sqlite3, glob
for d in ('pf_fetch','pf_write','pf_lds'):
    for path in glob.glob(f'gpurun_out/r6/{d}/**/*.db', recursive=True):
        db = sqlite3.connect(path)
        for k, c, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
            if 'gemm_f32' in k: print(d, c, n, f'{v:.4g}')
PY
rm -rf $OUT/pf_write $OUT/pf_sq $OUT/pf_lds $OUT/pf_fetch
grep "gemm_f32\|kernel " $OUT/f32_pmc_a.txt | cut -c1-260
grep "TF/s" $OUT/pf_sq.log
