#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for d in 0 1 2 3; do
  echo "ODIN_BP_PCDBG=$d $(ODIN_BP_PC=1 ODIN_BP_PCDBG=$d KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -E 'dec4|dec3' | tr '\n' ' ')"
done
