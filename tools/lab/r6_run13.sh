#!/bin/bash
# round 6, GPU call 13: residual add folded into the LayerNorm forward (training plans): parity + same-box A/B (MDT_FUSE_RES_LN=0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "fwd_res or ln_modulate" > $OUT/t13_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $OUT/t13_kernels.log
timeout 1500 python -m pytest tests/test_10_engine_gpu.py -x -q > $OUT/t13_engine.log 2>&1; echo "engine rc=$?"; tail -3 $OUT/t13_engine.log | cut -c1-200
timeout 1500 python -m pytest tests/test_40_full_batch_gpu.py -x -q > $OUT/t13_full.log 2>&1; echo "full rc=$?"; tail -3 $OUT/t13_full.log | cut -c1-200
for r in 1 2; do
timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b13_fuse_$r.json 2> $OUT/b13.err
MDT_FUSE_RES_LN=0 timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b13_old_$r.json 2> $OUT/b13.err
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/b13_128_fuse.json 2> $OUT/b13.err
MDT_FUSE_RES_LN=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/b13_128_old.json 2> $OUT/b13.err
python - <<'PY'
import json
for n in ('b13_fuse_1','b13_old_1','b13_fuse_2','b13_old_2','b13_128_fuse','b13_128_old'):
    try:
        l=[x for x in open(f'gpurun_out/r6/{n}.json') if x.startswith('{')]
        d=json.loads(l[-1]); e=d['roofline'].get('encoder',{})
        print(n, d['value'], d['ms_per_step'], 'nt8', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'enc', e.get('frac'), e.get('fwd_ms'), e.get('bwd_ms'), 'loss', d['mean_loss'])
    except Exception as ex: print(n, 'ERR', ex)
PY
