#!/bin/bash
# round 5, GPU call 3: overlap GEMM on the decoder shapes (epilogue-dominated)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 600 python tools/nt8o_bench.py --iters 3 --rounds 2 --decoder --skip-parity > $OUT/nt8o_bench3.txt 2>&1
echo "nt8o rc=$?" >> $OUT/nt8o_bench3.txt
tail -40 $OUT/nt8o_bench3.txt | cut -c1-400
