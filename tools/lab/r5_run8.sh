#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r5
mkdir -p $OUT
timeout 900 python -m pytest tests/test_10_engine_gpu.py -q -x > $OUT/engine_tests2.log 2>&1
tail -4 $OUT/engine_tests2.log
python bench.py --resolution 64 --micro-batch 256 --steps 3 --warmup 1 --no-cpu-baseline --no-sampler > $OUT/bench_512_b.json 2> $OUT/bench_512_b.err
cut -c1-400 $OUT/bench_512_b.json
python bench.py --no-cpu-baseline > $OUT/bench_default_b.json 2> $OUT/bench_default_b.err
cut -c1-300 $OUT/bench_default_b.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench_default_b.json'))
print({k:d[k] for k in d if k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('encoder'), d.get('sampler'))
PY
