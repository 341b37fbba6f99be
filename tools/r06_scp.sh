#!/bin/bash
# RGB first layer forward on two f16 planes: parity on the GPU, Shapes3D / CelebA / FactorVAE steps both ways in one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_ops.py tests/test_gpu_parity.py -q -m gpu -k "conv2d or exact_fp32 or shapes3d or celeba or full_batch" 2>&1 | tail -3
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch --profile-ops "$@" 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; grep "enc0:conv" /tmp/err.txt; }
for w in celeba_betatcvae_b512 shapes3d_vae_b256 factorvae_shapes3d_b256; do
  for i in 1 2 3; do
    echo "$w planes             $(ab --workload $w)"
    echo "$w --no-smallc-planes $(ab --workload $w --no-smallc-planes)"
  done
done
