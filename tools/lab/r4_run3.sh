#!/bin/bash
# round 4, third GPU call: CU-partition experiment (weight gradients beside the data-gradient chain), counter inventory
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -i -E "mall|dram|TCC_EA0_RD|TCC_EA0_WR|TCC_BUBBLE|HBM" $OUT/counters_list.txt | head -60 > $OUT/counters_mem.txt
timeout 600 python tools/cu_partition_bench.py > $OUT/cu_partition.txt 2>&1
echo "rc=$?" >> $OUT/cu_partition.txt
cat $OUT/cu_partition.txt; wc -l $OUT/counters_list.txt; head -40 $OUT/counters_mem.txt
