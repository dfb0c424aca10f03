#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -q -k "attention" > $OUT/attn_tests2.log 2>&1
tail -8 $OUT/attn_tests2.log
timeout 300 python tools/attn_bench.py > $OUT/attn_bench2.txt 2>&1
cat $OUT/attn_bench2.txt
