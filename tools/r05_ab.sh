#!/bin/bash
# same-call A/B of two builds of the library: the product .so against tools/diag/libodin_prev.so (ODIN_HIP_LIB)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['north_star_3ch']['ms_per_step'], d['roofline']['us_per_launch'])"
  ODIN_HIP_LIB=$PWD/tools/diag/libodin_prev.so python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'], d['north_star_3ch']['ms_per_step'], d['roofline']['us_per_launch'])"
done
