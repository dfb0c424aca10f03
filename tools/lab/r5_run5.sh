#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 300 python tools/nt_split_bench.py > $OUT/nt_split1.txt 2>&1
cat $OUT/nt_split1.txt
