#!/bin/bash
# round 4: L2-miss latency calibration only (tools/mall_probe.py) -- four reductions: HBM / Infinity Cache x all CUs / 32 CUs
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4; mkdir -p $OUT
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum -d $OUT/pmc_latcal -o p -- python tools/mall_probe.py > $OUT/pmc_latcal.log 2>&1
python tools/pmc_latency.py $OUT/pmc_latcal reduce > $OUT/pmc_latcal.txt 2>&1
rm -rf $OUT/pmc_latcal
cat $OUT/pmc_latcal.txt; tail -3 $OUT/pmc_latcal.log
