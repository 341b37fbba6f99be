#!/bin/bash
# round-5 first probe: default line, eager (no graph) step, timeline
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-fit > gpurun_out/r05p1_default.json 2> gpurun_out/r05p1_default.err
python bench.py --no-cpu-baseline --no-fit --no-north-star-3ch --no-graph > gpurun_out/r05p1_nograph.json 2> gpurun_out/r05p1_nograph.err
python bench.py --no-cpu-baseline --no-fit --no-north-star-3ch > gpurun_out/r05p1_default2.json 2>/dev/null
./tools/profile.sh r05p1_prof --no-north-star-3ch --no-fit > /dev/null 2>&1
tail -2 gpurun_out/r05p1_prof_timeline.txt
python - <<'PY'
import json
for f in ('r05p1_default', 'r05p1_nograph', 'r05p1_default2'):
  try:
    d = json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
  except Exception as e:
    print(f, 'ERR', e)
PY
