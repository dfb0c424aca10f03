#!/bin/bash
# Final measurement battery of a round (run through gpurun):  bash tools/final_measure.sh <tag>
# bench lines + rocprofv3 kernel-trace summaries for the benchmarked configuration, the per-GPU share of the 8-GPU run,
# the 512^2 configuration and the sampler; per-shape GEMM tables.  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r4final}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
trace() {  # name, header, command...
  local name=$1 hdr=$2; shift 2
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- "$@" > $OUT/kt_$name.log 2>&1
  local db=$(find $OUT/kt_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $OUT/kernel_stats_$name.txt "$hdr" > /dev/null
  rm -rf $OUT/kt_$name
}
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=12 > $OUT/gputests_final.log 2>&1
echo "suite rc=$?" >> $OUT/gputests_final.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/bench_b128.json 2> $OUT/bench_b128.err
python bench.py --resolution 64 --micro-batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-sampler > $OUT/bench_xl2_512.json 2> $OUT/bench_xl2_512.err
trace final "rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler (round 4 FINAL build; 5 optimizer steps + plan construction)" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler
trace b128 "rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 2 --global-batch 128 --no-cpu-baseline --no-kernel-events --no-sampler (round 4 FINAL; per-GPU share of the 8-GPU run; 12 optimizer steps + plan construction)" python bench.py --steps 10 --warmup 2 --global-batch 128 --no-cpu-baseline --no-kernel-events --no-sampler
trace 512 "rocprofv3 --kernel-trace -- python bench.py --resolution 64 --micro-batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sampler (round 4 FINAL; XL/2 at 512^2 latents, 2 optimizer steps of 4 micro-batches)" python bench.py --resolution 64 --micro-batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sampler
trace sampler "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 10 (round 4 FINAL; XL/2, batch 64 x 2 CFG, 4 + 10 Heun steps = 26 network evaluations)" python tools/sampler_profile.py 10
trace vae "rocprofv3 --kernel-trace -- python tools/vae_profile.py 64 3 (round 4 FINAL; VAE decode, batch 64, 1 warm-up + 3 decodes)" python tools/vae_profile.py 64 3
python tools/nt8_bench.py --iters 3 --rounds 2 --decoder > $OUT/nt8_bench.txt 2>&1
python tools/tn8_ab.py 131072 0,8 > $OUT/tn8_bench.txt 2>&1
python tools/nt8_sched.py --scheds 0,1029,37,69,133,101,229 --kloop-only > $OUT/nt8_kloop_decomposition.txt 2>&1
cut -c1-700 $OUT/bench_default.json; echo; cut -c1-300 $OUT/bench_b128.json; echo; cut -c1-300 $OUT/bench_xl2_512.json; echo
tail -3 $OUT/gputests_final.log; head -30 $OUT/kernel_stats_final.txt | cut -c1-150
