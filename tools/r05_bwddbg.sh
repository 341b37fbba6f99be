#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
: > gpurun_out/r05_bwddbg.txt
for d in 0 64 0 64; do
  echo "ODIN_BP_DBG=$d $(ODIN_BP_DBG=$d KB_WHICH=bwd KB_ONLY=dec4 timeout 300 python tools/kbench.py 2>&1 | grep dec4)" >> gpurun_out/r05_bwddbg.txt
done
cat gpurun_out/r05_bwddbg.txt
