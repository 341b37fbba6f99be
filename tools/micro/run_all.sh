#!/bin/bash
# builds and runs the MFMA / VALU overlap microbenchmarks on the GPU box; output -> gpurun_out/r02_micro.txt
cd "$(dirname "$0")"
out=../gpurun_out/r02_micro.txt
mkdir -p ../gpurun_out
: > $out
for f in mfma_fillers mfma_bf16_fillers mfma_bf16_mix mfma_two_waves mfma_shadow; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/$f $f.hip 2>/dev/null || { echo "build failed: $f" >> $out; continue; }
  echo "==== $f" >> $out
  timeout 120 /tmp/$f >> $out 2>&1
done
