#!/bin/bash
# whole-iteration timeline of the audio VAE step on the block-window kernels
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
w=speech_vae_b256
out=gpurun_out/r06p6_$w; mkdir -p $out
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload $w --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
python tools/timeline.py $out 20 1 > gpurun_out/r06p6_tl_$w.txt 2>&1
rm -rf $out
cat gpurun_out/r06p6_tl_$w.txt | cut -c1-110
