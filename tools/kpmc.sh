#!/bin/bash
# SQ counter passes over tools/kbench.py (the three dominant plane kernels stand-alone)
# usage: tools/kpmc.sh <tag>
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  out=$root/gpurun_out/${tag}_kpmc_$i
  mkdir -p $out
  (cd $root && KB_REPS=5 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o pmc -- python3 tools/kbench.py > $out.log 2>&1)
  (cd $root && python tools/pmc_summary.py $out > gpurun_out/${tag}_kpmc_$i.txt)
  tail -3 $out.log
  rm -rf $out
done
