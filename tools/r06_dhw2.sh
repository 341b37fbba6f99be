#!/bin/bash
# prefetch depth of dense_h: stand-alone times of both builds, step A/B against tools/diag/libodin_prev.so in one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "dense or layer_bwd or ranged" 2>&1 | tail -3 > gpurun_out/r06_dhw2_tests.txt
python tools/densebench.py 2>&1 | grep -v amdgpu.ids | grep "min_tiles=128" > gpurun_out/r06_dhw2_bench.txt
ODIN_HIP_LIB=tools/diag/libodin_prev.so python tools/densebench.py 2>&1 | grep -v amdgpu.ids | grep "min_tiles=128" > gpurun_out/r06_dhw2_bench_prev.txt
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
{
  for w in factorvae_shapes3d_b256 celeba_betatcvae_b512 mnist_dense_b128; do
    for i in 1 2 3; do
      echo "$w new  $(ab --workload $w)"
      echo "$w prev $(ODIN_HIP_LIB=tools/diag/libodin_prev.so ab --workload $w)"
    done
  done
} > gpurun_out/r06_dhw2_ab.txt 2>&1
cat gpurun_out/r06_dhw2_tests.txt; echo NEW; cat gpurun_out/r06_dhw2_bench.txt; echo PREV; cat gpurun_out/r06_dhw2_bench_prev.txt gpurun_out/r06_dhw2_ab.txt
