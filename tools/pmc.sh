#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass) of a short bench.py run
# usage: tools/pmc.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | cut -d' ' -f1)
  out=$root/gpurun_out/${tag}_pmc_$name
  mkdir -p $out
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out -o pmc -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $out.log 2>&1
  (cd $root && python tools/pmc_summary.py $out > gpurun_out/${tag}_pmc_$name.txt)
  rm -rf $out
done
