cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python tools/sampler_profile.py 50 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3/kts -o kt -- python $GRAFT_REPO_ROOT/tools/sampler_profile.py 10 > $GRAFT_REPO_ROOT/gpurun_out/r3/kts.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r3/kts -name "*.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/r3/kernel_stats_sampler_mid.txt "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 10 (XL/2, batch 64 x 2 CFG, 4 + 10 Heun steps = 26 network evaluations; round 3 mid)" | cut -c1-170
rm -rf gpurun_out/r3/kts
