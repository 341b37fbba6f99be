#!/bin/bash
# round-5 probe 5: activation range words (forward guard): full GPU tests, A/B of the step with / without the words
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -6 > gpurun_out/r05p5_gpu_tests.txt
cat gpurun_out/r05p5_gpu_tests.txt
for i in 1 2 3; do
python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit > gpurun_out/r05p5_act_$i.json 2>/dev/null
ODIN_ACT_WORDS=0 python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit > gpurun_out/r05p5_noact_$i.json 2>/dev/null
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512 mnist_dense_b128; do
python bench.py --workload $w --no-cpu-baseline --no-north-star-3ch > gpurun_out/r05p5_act_$w.json 2>/dev/null
ODIN_ACT_WORDS=0 python bench.py --workload $w --no-cpu-baseline --no-north-star-3ch > gpurun_out/r05p5_noact_$w.json 2>/dev/null
done
./tools/profile.sh r05p5_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r05p5_prof_timeline.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05p5_*act_*.json')):
  try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    ns = d.get('north_star_3ch') or {}
    print(f.split('/')[-1], d['value'], d['ms_per_step'], ns.get('ms_per_step'))
  except Exception as e:
    print(f, 'ERR', e)
PY
