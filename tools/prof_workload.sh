#!/bin/bash
# usage: prof_w.sh <tag> <workload>
tag=$1; w=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $root/bench.py --no-cpu-baseline --no-north-star-3ch --workload $w --steps 30 --warmup 5 > $out.log 2>&1
cd $root
python tools/prof_summary.py $out > gpurun_out/${tag}_summary.txt
rm -rf $out
