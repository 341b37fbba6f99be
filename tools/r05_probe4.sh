#!/bin/bash
# round-5 probe 4: device-resident hyper-parameter ring A/B (same box, same call), tests touching the update path
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api.py tests/test_golden_vae.py -m gpu -x -q -n 2 2>&1 | tail -4 > gpurun_out/r05p4_gpu_tests.txt
cat gpurun_out/r05p4_gpu_tests.txt
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-exact-fp32 --no-north-star-3ch > gpurun_out/r05p4_ring_$i.json 2>/dev/null
ODIN_HYPER_RING=0 python bench.py --no-cpu-baseline --no-exact-fp32 --no-north-star-3ch > gpurun_out/r05p4_copy_$i.json 2>/dev/null
done
./tools/profile.sh r05p4_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
tail -8 gpurun_out/r05p4_prof_timeline.txt
python - <<'PY'
import json
for f in ('r05p4_ring_1', 'r05p4_copy_1', 'r05p4_ring_2', 'r05p4_copy_2', 'r05p4_ring_3', 'r05p4_copy_3'):
  try:
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], {k: v['frac_of_step_replay'] for k, v in d.get('fit', {}).items()})
  except Exception as e:
    print(f, 'ERR', e)
PY
