#!/bin/bash
# fused Conv2DTranspose backward (bwd_planes.hip): parity cases, stand-alone timing against the two launches, step A/B
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -x -k "layer_bwd_in_one_call" 2>&1 | tail -3 > gpurun_out/r05_bwdprobe.txt
KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_bwdprobe.txt
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
ODIN_BP_NSETS=2 KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/NSETS=2 /' >> gpurun_out/r05_bwdprobe.txt
for i in 1 2; do
  echo "fused bwd            $(ab)" >> gpurun_out/r05_bwdprobe.txt
  echo "ODIN_BP_NSETS=2      $(ODIN_BP_NSETS=2 ab)" >> gpurun_out/r05_bwdprobe.txt
  echo "ODIN_NOBWDPLANES=1   $(ODIN_NOBWDPLANES=1 ab)" >> gpurun_out/r05_bwdprobe.txt
done
cat gpurun_out/r05_bwdprobe.txt
