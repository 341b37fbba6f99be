#!/bin/bash
# fixed cost of a plane-kernel launch: the dec4 / dec3 backward at batch sizes that leave 1, 2, 4, 8, 32 tiles per workgroup
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for b in 8 16 32 64 256; do
  echo "B=$b 8-wave: $(KB_B=$b KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -E 'dec4|dec3' | sed 's/wgrad + dgrad, //' | tr '\n' ' ')"
  echo "B=$b pc:     $(ODIN_BP_PC=1 KB_B=$b KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -E 'dec4|dec3' | sed 's/wgrad + dgrad, //' | tr '\n' ' ')"
done
