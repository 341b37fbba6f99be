#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
echo "pc:     $(KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -E 'dec' | sed 's/wgrad + dgrad, //; s/two launches: [0-9.]* us; //' | tr '\n' ' ')"
echo "8-wave: $(ODIN_HIP_LIB=$PWD/tools/diag/libodin_prev.so KB_WHICH=bwd timeout 300 python tools/kbench.py 2>&1 | grep -E 'dec' | sed 's/wgrad + dgrad, //; s/two launches: [0-9.]* us; //' | tr '\n' ' ')"
done
