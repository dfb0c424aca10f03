#!/bin/bash
# round 6, GPU call 4: gemm_tn8 long + tail partition -- parity tests, A/B against uniform splits (tn8_dbg bit 8), bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "tn8 or gemm_tn or poisoned" > $OUT/t4_kernels.log 2>&1; echo "kernels rc=$?"
tail -3 $OUT/t4_kernels.log | cut -c1-200
timeout 600 python tools/tn8_ab.py 131072 0,256 > $OUT/tn8_ab_131072.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_ab_131072.txt | cut -c1-200
timeout 600 python tools/tn8_ab.py 16384 0,256 > $OUT/tn8_ab_16384.txt 2>&1; grep -v amdgpu.ids $OUT/tn8_ab_16384.txt | cut -c1-200
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b4.json 2> $OUT/b4.err; echo "bench rc=$?"
MDT_TUNE=tn8_dbg=256 timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b4_uniform.json 2> $OUT/b4_uniform.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b4b.json 2> $OUT/b4b.err; echo "bench rc=$?"
MDT_TUNE=tn8_dbg=256 timeout 900 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-sampler > $OUT/b4b_uniform.json 2> $OUT/b4b_uniform.err; echo "bench rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/b4_128.json 2> $OUT/b4_128.err
MDT_TUNE=tn8_dbg=256 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sampler --global-batch 128 > $OUT/b4_128_uniform.json 2> $OUT/b4_128_uniform.err
python - <<'PY'
import json
for n in ('b4','b4_uniform','b4b','b4b_uniform','b4_128','b4_128_uniform'):
    try:
        l=[x for x in open(f'gpurun_out/r6/{n}.json') if x.startswith('{')]
        d=json.loads(l[-1]); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('encoder',{}).get('frac'))
    except Exception as e: print(n, 'ERR', e)
PY
