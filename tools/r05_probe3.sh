#!/bin/bash
# round-5 probe 3: latent_block2 (bottleneck + first deconv in one launch): tests, A/B, timeline
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pointwise.py tests/test_gpu_parity.py tests/test_api.py -m gpu -x -q -n 2 2>&1 | tail -6 > gpurun_out/r05p3_gpu_tests.txt
cat gpurun_out/r05p3_gpu_tests.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 > gpurun_out/r05p3_v2_$i.json 2>/dev/null
ODIN_LATBLOCK2=0 python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 > gpurun_out/r05p3_v1_$i.json 2>/dev/null
done
./tools/profile.sh r05p3_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r05p3_prof_timeline.txt
python - <<'PY'
import json
for f in ('r05p3_v2_1', 'r05p3_v1_1', 'r05p3_v2_2', 'r05p3_v1_2'):
  try:
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['north_star_3ch']['ms_per_step'])
  except Exception as e:
    print(f, 'ERR', e)
PY
