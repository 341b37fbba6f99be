#!/bin/bash
# usage: tools/regs.sh <kernel file stem> [extra hipcc flags]: per-kernel register / spill report of one translation unit
stem=$1; shift
cd /root/repo/odin_ai_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c -o /tmp/$stem.o $stem.hip -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|Spill|ScratchSize|SGPRs:|LDS Size" | sed 's/.*remark: *//;s/\[-Rpass.*//' \
  | awk '/Function Name/{printf "\n%s ", $3} !/Function Name/{printf "%s ", $0}'; echo
