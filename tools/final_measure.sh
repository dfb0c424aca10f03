#!/bin/bash
# Final measurement battery of a round (run through gpurun):  bash tools/final_measure.sh <tag> [round label]
# GPU test suite, bench lines (default, --no-kernel-events, per-GPU share of the 8-GPU run, 512^2 shapes), rocprofv3 kernel-trace
# summaries of the same commands, per-shape GEMM tables.  Everything lands in gpurun_out/<tag>/; every text artefact starts
# with the kernel-source hash of the build it was taken on (maskdit_amd._lib.source_hash()).
TAG=${1:-r6final}
ROUND=${2:-round 6 FINAL build}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
# evidence called FINAL must be of the committed tree.  The GPU box has no .git, so the LOCAL caller
# (tools/final_local.sh) refuses a dirty tree and writes the commit + kernel-source hash it verified into
# gpurun_out/.final_stamp, which travels here with the snapshot; this script refuses to run if the sources it finds do not
# hash to that value.
HASH=$(python -c "from maskdit_amd import _lib; print(_lib.source_hash())")
if [ ! -f tools/.final_stamp ]; then echo "final_measure: tools/.final_stamp missing -- run it through tools/final_local.sh (refuses a dirty tree)"; exit 3; fi
WANT=$(cut -d' ' -f2 tools/.final_stamp)
COMMIT=$(cut -d' ' -f1 tools/.final_stamp)
if [ "$WANT" != "$HASH" ]; then echo "final_measure: kernel sources hash to $HASH, the stamp of commit $COMMIT says $WANT -- refusing"; exit 3; fi
ROUND="$ROUND, commit $COMMIT"
stamp() {  # file: prepend the hash line
  local f=$1
  { echo "# kernel-source hash $HASH ($ROUND)"; cat $f; } > $f.tmp && mv $f.tmp $f
}
trace() {  # name, header, command...
  local name=$1 hdr=$2; shift 2
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- "$@" > $OUT/kt_$name.log 2>&1
  local db=$(find $OUT/kt_$name -name "*.db" | head -1)
  python tools/rocprof_summary.py $db $OUT/kernel_stats_$name.txt "$hdr" > /dev/null
  stamp $OUT/kernel_stats_$name.txt
  rm -rf $OUT/kt_$name
}
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=12 > $OUT/gputests_final.log 2>&1
echo "suite rc=$?" >> $OUT/gputests_final.log
stamp $OUT/gputests_final.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --no-kernel-events --no-cpu-baseline --no-sampler > $OUT/bench_default_no_kernel_events.json 2> $OUT/bench_default_no_kernel_events.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/bench_b128.json 2> $OUT/bench_b128.err
python bench.py --resolution 64 --micro-batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-sampler > $OUT/bench_xl2_512.json 2> $OUT/bench_xl2_512.err
trace final "rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler ($ROUND; 5 optimizer steps + plan construction)" python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sampler
trace b128 "rocprofv3 --kernel-trace -- python bench.py --steps 10 --warmup 2 --global-batch 128 --no-cpu-baseline --no-kernel-events --no-sampler ($ROUND; per-GPU share of the 8-GPU run; 12 optimizer steps + plan construction)" python bench.py --steps 10 --warmup 2 --global-batch 128 --no-cpu-baseline --no-kernel-events --no-sampler
trace 512 "rocprofv3 --kernel-trace -- python bench.py --resolution 64 --micro-batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sampler ($ROUND; XL/2 at 512^2 latents, 2 optimizer steps of 4 micro-batches)" python bench.py --resolution 64 --micro-batch 256 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events --no-sampler
trace sampler "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 10 ($ROUND; XL/2, batch 64 x 2 CFG, 4 + 10 Heun steps = 26 network evaluations)" python tools/sampler_profile.py 10
trace vae "rocprofv3 --kernel-trace -- python tools/vae_profile.py 64 3 ($ROUND; VAE decode, batch 64, 1 warm-up + 3 decodes)" python tools/vae_profile.py 64 3
trace sampler_fp32 "rocprofv3 --kernel-trace -- python tools/sampler_profile.py 3 fp32 ($ROUND; XL/2, batch 64 x 2 CFG, exact-fp32 plan, 2 + 3 Heun steps = 8 network evaluations)" python tools/sampler_profile.py 3 fp32
python tools/f32_bench.py > $OUT/f32_bench.txt 2>&1; stamp $OUT/f32_bench.txt
python tools/nt8_bench.py --iters 3 --rounds 2 --decoder > $OUT/nt8_bench.txt 2>&1; stamp $OUT/nt8_bench.txt
python tools/tn8_ab.py 131072 0,256 > $OUT/tn8_bench.txt 2>&1; stamp $OUT/tn8_bench.txt
python tools/nt8o_bench.py --iters 3 --rounds 2 --decoder > $OUT/nt8o_bench.txt 2>&1; stamp $OUT/nt8o_bench.txt
python tools/nt_split_bench.py > $OUT/nt_forms_b128.txt 2>&1; stamp $OUT/nt_forms_b128.txt
cut -c1-700 $OUT/bench_default.json; echo; cut -c1-300 $OUT/bench_default_no_kernel_events.json; echo; cut -c1-300 $OUT/bench_b128.json; echo; cut -c1-300 $OUT/bench_xl2_512.json; echo
tail -4 $OUT/gputests_final.log; head -30 $OUT/kernel_stats_final.txt | cut -c1-150
