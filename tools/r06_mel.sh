#!/bin/bash
# stft_mel512_kernel: parity on the GPU, the front-end launch alone and the speech step, both ways, one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_mel.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06_mel_tests.txt
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch --workload speech_vae_b256 --profile-ops "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('mel_kernel'))"; }
{
  for i in 1 2 3; do
    echo "speech r16          $(ab)"
    echo "speech --no-mel-r16 $(ab --no-mel-r16)"
  done
} > gpurun_out/r06_mel_ab.txt 2>&1
cat gpurun_out/r06_mel_tests.txt gpurun_out/r06_mel_ab.txt
