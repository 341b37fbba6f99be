#!/bin/bash
# GPU suite + default bench + the other workloads (short)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -n 2 2>&1 | tail -4 > gpurun_out/r05_check.txt
one() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
echo "default $(one)" >> gpurun_out/r05_check.txt
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  echo "$w $(one --workload $w --no-north-star-3ch)" >> gpurun_out/r05_check.txt
done
cat gpurun_out/r05_check.txt
