#!/bin/bash
# producer/consumer form of the fused Conv2DTranspose backward (ODIN_BP_PC=1) against the 8-wave form
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
ODIN_BP_PC=1 timeout 600 python -m pytest tests/test_ops.py -m gpu -q -x -k "layer_bwd_in_one_call" 2>&1 | tail -3 > gpurun_out/r05_pcprobe.txt
KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_pcprobe.txt
ODIN_BP_PC=1 KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/PC /' >> gpurun_out/r05_pcprobe.txt
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
for i in 1 2; do
  echo "8-wave          $(ab)" >> gpurun_out/r05_pcprobe.txt
  echo "ODIN_BP_PC=1    $(ODIN_BP_PC=1 ab)" >> gpurun_out/r05_pcprobe.txt
done
cat gpurun_out/r05_pcprobe.txt
