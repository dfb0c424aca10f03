#!/bin/bash
# round 4: the batch-1024 step against the fp32 CPU ORACLE (MASKDIT_SLOW=1: ~6 min of host work) on the final build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4; mkdir -p $OUT
python -c "from maskdit_amd import _lib; print('kernel-source hash', _lib.source_hash())" > $OUT/full_batch_vs_oracle.txt
MASKDIT_SLOW=1 timeout 1500 python -m pytest tests/test_40_full_batch_gpu.py -m gpu -q -s -k "vs_oracle_slices" >> $OUT/full_batch_vs_oracle.txt 2>&1
echo "rc=$?" >> $OUT/full_batch_vs_oracle.txt
tail -25 $OUT/full_batch_vs_oracle.txt
