#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/r06p7_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06p7_tests.txt
ab() { python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('north_star_3ch') or {}).get('ms_per_step'))"; }
for i in 1 2; do
  echo "dsprites           $(ab)" >> gpurun_out/r06p7_ab.txt
  echo "dsprites no neck   $(ab --engine-opt neck=False --no-north-star-3ch)" >> gpurun_out/r06p7_ab.txt
  echo "factorvae          $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch)" >> gpurun_out/r06p7_ab.txt
  echo "factorvae no neck  $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch --engine-opt neck=False)" >> gpurun_out/r06p7_ab.txt
done
for w in factorvae_shapes3d_b256; do
  out=gpurun_out/r06p7_$w; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload $w --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
  python tools/timeline.py $out 20 2 > gpurun_out/r06p7_tl_$w.txt 2>&1
  rm -rf $out
done
./tools/profile.sh r06p7_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r06p7_tests.txt gpurun_out/r06p7_ab.txt
