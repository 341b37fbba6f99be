#!/bin/bash
# neck probe: GPU tests, default line with and without the neck, one-step timeline, FactorVAE
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/r06p5_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06p5_tests.txt
ab() { python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('north_star_3ch') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  echo "neck    $(ab)" >> gpurun_out/r06p5_ab.txt
  echo "no neck $(ab --engine-opt neck=False --no-north-star-3ch)" >> gpurun_out/r06p5_ab.txt
done
echo "factorvae neck    $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch)" >> gpurun_out/r06p5_ab.txt
echo "factorvae no neck $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch --engine-opt neck=False)" >> gpurun_out/r06p5_ab.txt
./tools/profile.sh r06p5_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r06p5_tests.txt gpurun_out/r06p5_ab.txt; cat gpurun_out/r06p5_prof_timeline.txt | cut -c1-120
