#!/bin/bash
# round-5 probe 6: smalldeconv (the decoders' first Conv2DTranspose as its own 1024-thread launches): tests, timeline, step
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 > gpurun_out/r05p6_gpu_tests.txt
cat gpurun_out/r05p6_gpu_tests.txt
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit > gpurun_out/r05p6_$i.json 2>/dev/null
done
./tools/profile.sh r05p6_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r05p6_prof_timeline.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05p6_?.json')):
  d = json.loads(open(f).read().strip().splitlines()[-1])
  print(f.split('/')[-1], d['value'], d['ms_per_step'], (d.get('north_star_3ch') or {}).get('ms_per_step'))
PY
