#!/bin/bash
# round 5, GPU call 1: first run of the wave-specialised overlap GEMM (parity + timing decomposition) and the new
# reference-fixture test of the batch-1024 step
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 600 python tools/nt8o_bench.py --iters 3 --rounds 2 > $OUT/nt8o_bench1.txt 2>&1
echo "nt8o rc=$?" >> $OUT/nt8o_bench1.txt
tail -60 $OUT/nt8o_bench1.txt
timeout 420 python -m pytest tests/test_40_full_batch_gpu.py -q -s -k reference_fixture > $OUT/bs1024_fixture.log 2>&1
echo "fixture rc=$?" >> $OUT/bs1024_fixture.log
tail -15 $OUT/bs1024_fixture.log
