#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
HASH=$(python -c "from maskdit_amd import _lib; print(_lib.source_hash())")
timeout 600 python tools/nt8o_bench.py --iters 3 --rounds 2 --decoder > $OUT/nt8o_bench_final.txt 2>&1
{ echo "# kernel-source hash $HASH (round 5 FINAL kernel sources)"; cat $OUT/nt8o_bench_final.txt; } > $OUT/nt8o_bench_final.txt.tmp && mv $OUT/nt8o_bench_final.txt.tmp $OUT/nt8o_bench_final.txt
grep -v "^parity (" $OUT/nt8o_bench_final.txt | head -30 | cut -c1-260
bash tools/pmc_refresh.sh r5pmc2 > $OUT/pmc2.log 2>&1
cat gpurun_out/r5pmc2/pmc_gemm_nt.json | head -12
