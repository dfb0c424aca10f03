#!/bin/bash
# round 4, fifth GPU call: VAE decode with the fused convolution epilogue (skip connection + GroupNorm statistics)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4
mkdir -p $OUT
timeout 900 python -m pytest tests/test_01_vae_gpu.py tests/test_30_entry_gpu.py -m gpu -q -k "vae or conv3x3 or gn_stats or softmax" -s > $OUT/vae_tests.log 2>&1
echo "vae tests rc=$?" >> $OUT/vae_tests.log
rm -f $OUT/vae_ab.txt
for i in 1 2; do
  for c in 128 256 512; do echo "MDT_VAE_FUSE_MAXC=$c" >> $OUT/vae_ab.txt; MDT_VAE_FUSE_MAXC=$c timeout 300 python tools/vae_profile.py 64 5 >> $OUT/vae_ab.txt 2>&1; done
  MDT_VAE_FUSE=0 timeout 300 python tools/vae_profile.py 64 5 >> $OUT/vae_ab.txt 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/kt_vae -o kt -- python tools/vae_profile.py 64 3 > $OUT/kt_vae.log 2>&1
DB=$(find $OUT/kt_vae -name "*.db" | head -1)
python tools/rocprof_summary.py $DB $OUT/kernel_stats_vae.txt "rocprofv3 --kernel-trace -- python tools/vae_profile.py 64 3 (round 4: fused skip-connection + GroupNorm-statistics epilogue; 1 warm-up + 3 decodes of batch 64)" > /dev/null
rm -rf $OUT/kt_vae
grep -E "passed|failed|rc=|VAE decode|fused Group|conv3x3" $OUT/vae_tests.log | tail -20; grep -v amdgpu.ids $OUT/vae_ab.txt; head -16 $OUT/kernel_stats_vae.txt | cut -c1-140
