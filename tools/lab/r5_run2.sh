#!/bin/bash
# round 5, GPU call 2: overlap GEMM with 2 / 3 / 4 loader waves
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 600 python tools/nt8o_bench.py --iters 3 --rounds 2 > $OUT/nt8o_bench2.txt 2>&1
echo "nt8o rc=$?" >> $OUT/nt8o_bench2.txt
grep -v "^parity (" $OUT/nt8o_bench2.txt | tail -60
grep -c "OK$" $OUT/nt8o_bench2.txt; grep "FAIL" $OUT/nt8o_bench2.txt | head
