#!/bin/bash
# round-6 first look: the changed paths' GPU tests, the default line on this box, whole-iteration timelines of the
# workloads the review names (FactorVAE incl. its discriminator half, speech, MNIST conv)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_api.py -m gpu -x -q -n 2 2>&1 | tail -5 > gpurun_out/r06p1_tests.txt
python bench.py --no-cpu-baseline --no-fit > gpurun_out/r06p1_bench.json 2> gpurun_out/r06p1_bench.err
for w in factorvae_shapes3d_b256 speech_vae_b256 mnist_conv_b128; do
  out=gpurun_out/r06p1_$w; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload $w --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
  per=1; [ $w = factorvae_shapes3d_b256 ] && per=2
  python tools/timeline.py $out 20 $per > gpurun_out/r06p1_tl_$w.txt 2>&1
  python tools/prof_summary.py $out > gpurun_out/r06p1_sum_$w.txt 2>&1
  rm -rf $out
done
tail -3 gpurun_out/r06p1_tests.txt; cat gpurun_out/r06p1_bench.json | cut -c1-400
