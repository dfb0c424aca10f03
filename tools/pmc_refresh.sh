#!/bin/bash
# Counter evidence for the CURRENT build (run on the GPU box through gpurun):  bash tools/pmc_refresh.sh <tag>
# Separate rocprofv3 --pmc passes of the benchmarked command (TCC: FETCH_SIZE needs 3 of the 4 slots; never combined with
# trace domains other than the kernel trace), summaries written under gpurun_out/<tag>/ for copying into profiles/.
set -u
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python bench.py --steps 1 --warmup 1 --micro-batch 1024 --no-cpu-baseline --no-sampler --no-kernel-events"
run_pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o p -- $CMD > $OUT/pmc_$name.log 2>&1
}
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS GRBM_GUI_ACTIVE
F=$(find $OUT/pmc_fetch -name "*.db" | head -1)
W=$(find $OUT/pmc_write -name "*.db" | head -1)
python tools/pmc_summary.py $F $W $OUT/pmc_gemm_nt.json $OUT/pmc_hbm_traffic.txt DiT-XL/2 32 1024 "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- $CMD (separate passes)" > /dev/null
python tools/pmc_table.py $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_lds > $OUT/pmc_counters.txt 2>&1
# L2-miss latency (tools/mall_probe.py): the benchmarked step and the HBM / Infinity-Cache calibration kernels
run_pass lat TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum -d $OUT/pmc_latcal -o p -- python tools/mall_probe.py > $OUT/pmc_latcal.log 2>&1
{ echo "# mean L2-miss latency per kernel: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum (separate pass)"; echo "## calibration (tools/mall_probe.py): 3 passes over 8 GiB = HBM; 40 passes over 96 MiB = L2 misses served by the Infinity Cache"; python tools/pmc_latency.py $OUT/pmc_latcal reduce; echo; echo "## $CMD"; python tools/pmc_latency.py $OUT/pmc_lat; } > $OUT/pmc_miss_latency.txt 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_lds $OUT/pmc_lat $OUT/pmc_latcal
head -30 $OUT/pmc_hbm_traffic.txt | cut -c1-150
head -60 $OUT/pmc_counters.txt | cut -c1-250
head -40 $OUT/pmc_miss_latency.txt | cut -c1-160
