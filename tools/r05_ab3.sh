#!/bin/bash
# same-call A/B of two builds (product .so vs tools/diag/libodin_prev.so), both with the environment given as arguments
cd ${GRAFT_REPO_ROOT:-$(pwd)}
ab() { env "$@" python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['north_star_3ch']['ms_per_step'])"; }
for i in 1 2 3; do
  echo "new  $(ab "$@")"
  echo "prev $(ab "$@" ODIN_HIP_LIB=$PWD/tools/diag/libodin_prev.so)"
done
