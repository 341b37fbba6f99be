#!/bin/bash
# same-call A/B of environment switches: tools/r05_ab2.sh "VAR=1" ["VAR2=x" ...]   (each against the default, 3 rounds)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
: > gpurun_out/r05_ab2.txt
for i in 1 2 3; do
  echo "default   $(ab)" >> gpurun_out/r05_ab2.txt
  for v in "$@"; do
    echo "$v   $(env $v bash -c "$(declare -f ab); ab")" >> gpurun_out/r05_ab2.txt
  done
done
cat gpurun_out/r05_ab2.txt
