#!/bin/bash
# round-5 probe 2: GPU tests after the range-contract change, default line with exact_fp32, hyper-copy experiment
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -8 > gpurun_out/r05p2_gpu_tests.txt
python bench.py --no-cpu-baseline --no-fit > gpurun_out/r05p2_default.json 2> gpurun_out/r05p2_default.err
ODIN_SKIP_HYPER_COPY=1 python bench.py --no-cpu-baseline --no-fit --no-north-star-3ch --no-exact-fp32 > gpurun_out/r05p2_nohyper.json 2>/dev/null
python bench.py --no-cpu-baseline --no-fit --no-north-star-3ch --no-exact-fp32 > gpurun_out/r05p2_default2.json 2>/dev/null
cat gpurun_out/r05p2_gpu_tests.txt
python - <<'PY'
import json
for f in ('r05p2_default', 'r05p2_nohyper', 'r05p2_default2'):
  try:
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('exact_fp32'))
  except Exception as e:
    print(f, 'ERR', e)
PY
