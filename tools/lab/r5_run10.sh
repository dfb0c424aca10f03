#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
HASH=$(python -c "from maskdit_amd import _lib; print(_lib.source_hash())")
timeout 600 python -m pytest tests/test_00_kernels_gpu.py -q -k "nt8o" > $OUT/nt8o_tests3.log 2>&1; tail -3 $OUT/nt8o_tests3.log
timeout 600 python tools/nt8o_bench.py --iters 3 --rounds 2 --decoder > $OUT/nt8o_bench_final.txt 2>&1
{ echo "# kernel-source hash $HASH (round 5 FINAL kernel sources)"; cat $OUT/nt8o_bench_final.txt; } > $OUT/nt8o_bench_final.txt.tmp && mv $OUT/nt8o_bench_final.txt.tmp $OUT/nt8o_bench_final.txt
grep -v "^parity (" $OUT/nt8o_bench_final.txt | head -8 | cut -c1-200; grep -c "OK$" $OUT/nt8o_bench_final.txt; grep "FAIL\|ABORT" $OUT/nt8o_bench_final.txt | head
bash tools/pmc_refresh.sh r5pmc3 > $OUT/pmc3.log 2>&1
grep source_hash gpurun_out/r5pmc3/pmc_gemm_nt.json
