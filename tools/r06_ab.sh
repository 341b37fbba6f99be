#!/bin/bash
# same-call A/B of engine options: tools/r06_ab.sh <out tag> "<optA>" "<optB>" [rounds] [workload]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; A=$2; Bo=$3; n=${4:-3}; w=${5:-dsprites_betavae_b256}
ab() { python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 --no-north-star-3ch --workload $w "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in $(seq $n); do
  echo "$w [$A] $(ab $A)" >> gpurun_out/$tag.txt
  echo "$w [$Bo] $(ab $Bo)" >> gpurun_out/$tag.txt
done
cat gpurun_out/$tag.txt
