#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
rm -f gpurun_out/r06p8_ab.txt
timeout 900 python -m pytest tests/test_ops.py tests/test_neck.py tests/test_api.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06p8_tests.txt
ab() { python bench.py --no-cpu-baseline --no-fit --no-exact-fp32 --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do
  for w in shapes3d_vae_b256 factorvae_shapes3d_b256 dsprites_betavae_b256; do
    echo "$w neck_bwd=True  $(ab --workload $w --engine-opt neck_bwd=True)" >> gpurun_out/r06p8_ab.txt
    echo "$w neck_bwd=False $(ab --workload $w --engine-opt neck_bwd=False)" >> gpurun_out/r06p8_ab.txt
  done
done
out=gpurun_out/r06p8_fv; mkdir -p $out
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload factorvae_shapes3d_b256 --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
python tools/timeline.py $out 20 2 | grep -i "thin\|think\|mean\|dtc\|perm" > gpurun_out/r06p8_tl_fv.txt 2>&1
rm -rf $out
cat gpurun_out/r06p8_tests.txt gpurun_out/r06p8_ab.txt gpurun_out/r06p8_tl_fv.txt
