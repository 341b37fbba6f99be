#!/bin/bash
# samples per workgroup of smalldeconv.hip (needs a build that reads ODIN_SD_S in sd_samples(): a temporary switch of
# round 5, removed after this sweep -- kept as the record of how the value was chosen)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for S in 1 2 4; do
  export ODIN_SD_S=$S
  python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S', os.environ['ODIN_SD_S'], d['ms_per_step'], d['north_star_3ch']['ms_per_step'])"
done
