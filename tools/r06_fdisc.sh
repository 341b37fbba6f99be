#!/bin/bash
# fused discriminator head / permute_dims / folded Adam: parity on the GPU, FactorVAE iteration A/B in one call, timeline
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops.py tests/test_pointwise.py tests/test_gpu_parity.py tests/test_neck.py -q -m gpu -k "head or permute or factor or neck or folded" 2>&1 | tail -3 > gpurun_out/r06_fdisc_tests.txt
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
  for i in 1 2 3; do
    echo "factorvae fused           $(ab --workload factorvae_shapes3d_b256)"
    echo "factorvae --no-fused-disc $(ab --workload factorvae_shapes3d_b256 --no-fused-disc)"
  done
} > gpurun_out/r06_fdisc_ab.txt 2>&1
out=gpurun_out/r06_fdisc_tl; mkdir -p $out
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload factorvae_shapes3d_b256 --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
python tools/timeline.py $out 20 2 > gpurun_out/r06_fdisc_timeline.txt 2>&1
rm -rf $out
cat gpurun_out/r06_fdisc_tests.txt gpurun_out/r06_fdisc_ab.txt; tail -16 gpurun_out/r06_fdisc_timeline.txt
