#!/bin/bash
# round 6, GPU call 1: first run of the fp32-faithful path + the kernels the ADVICE fixes touched
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_00_kernels_gpu.py -x -q -k "f32 or nt8o or padded_keys" -s > $OUT/t1_kernels.log 2>&1; echo "kernels rc=$?"
tail -15 $OUT/t1_kernels.log | cut -c1-220
timeout 1200 python -m pytest tests/test_10_engine_gpu.py -x -q -k "fp32 or sampler" -s > $OUT/t1_engine.log 2>&1; echo "engine rc=$?"
grep -i "fp32\|passed\|failed\|error" $OUT/t1_engine.log | tail -20 | cut -c1-260
timeout 900 python -m pytest tests/test_20_ddp_gpu.py -x -q > $OUT/t1_ddp.log 2>&1; echo "ddp rc=$?"
tail -3 $OUT/t1_ddp.log | cut -c1-220
timeout 900 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/b1.json 2> $OUT/b1.err; echo "bench rc=$?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6/b1.json') if x.startswith('{')]
d=json.loads(l[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], json.dumps(d['sampler'])[:900])
PY
