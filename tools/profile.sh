#!/bin/bash
# rocprofv3 kernel trace of one bench.py run -> gpurun_out/<tag>_summary.txt (+ raw stats csv)
# usage: tools/profile.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $root/bench.py --no-cpu-baseline "$@" > $out.log 2>&1
cd $root
python tools/prof_summary.py $out > gpurun_out/${tag}_summary.txt
find $out -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
python tools/timeline.py $out > gpurun_out/${tag}_timeline.txt 2>&1
rm -rf $out
