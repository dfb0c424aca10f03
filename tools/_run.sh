cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python tools/nt8_sched.py --scheds 0,261,100 > gpurun_out/r3/sched4.log 2>&1
cat gpurun_out/r3/sched4.log
