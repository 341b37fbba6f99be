#!/bin/bash
# dense_hd: parity on the GPU, stand-alone times, CelebA / MNIST dense / FactorVAE steps both ways in one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python -m pytest tests/test_ops.py -q -m gpu -k "dense or layer_bwd or ranged" 2>&1 | tail -3
python tools/densebench.py 2>&1 | grep -v amdgpu.ids
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for w in celeba_betatcvae_b512 mnist_dense_b128; do
  for i in 1 2 3; do
    echo "$w dense_hw/hd   $(ab --workload $w)"
    echo "$w --no-dense-hw $(ab --workload $w --no-dense-hw)"
  done
done
