#!/bin/bash
# round-3 evidence run on the GPU box: tests, default bench with per-op table, rocprofv3 kernel trace + timeline,
# FETCH / WRITE / SQ PMC passes -> measured HBM traffic of the dominant kernels (fails when a priced kernel no
# longer exists), plane-kernel SQ counters, in-kernel stamps of the small-layer kernels, the other workloads, the
# RCCL path at world size 1.  Everything lands in gpurun_out/; copy what is to be judged into profiles/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
set -o pipefail
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r03_final_gpu_tests.txt
python bench.py --profile-ops > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err
grep "^#" gpurun_out/r03_final_bench.err > gpurun_out/r03_final_per_op.txt
./tools/profile.sh r03_final_prof --no-north-star-3ch > /dev/null 2>&1
./tools/pmc.sh r03 --no-north-star-3ch > /dev/null 2>&1
python tools/pmc_traffic.py r03 > /dev/null 2> gpurun_out/r03_final_pmc_traffic.err || echo "PMC TRAFFIC TABLE STALE" >> gpurun_out/r03_final_pmc_traffic.err
python bench.py --no-cpu-baseline > gpurun_out/r03_final_bench2.json 2>/dev/null   # picks up the fresh traffic file
./tools/kpmc.sh r03f > /dev/null 2>&1
timeout 200 python tools/stamps_ig.py > gpurun_out/r03_final_stamps_igemm.txt 2>&1
python tools/igbench.py > gpurun_out/r03_final_igbench.txt 2>&1
ODIN_NOIGEMM=1 python tools/igbench.py > gpurun_out/r03_final_igbench_tiled.txt 2>&1
python tools/enc0bench.py > gpurun_out/r03_final_enc0bench.txt 2>&1
timeout 200 python tools/stamps_fp.py > gpurun_out/r03_final_stamps_fconv_planes.txt 2>&1
timeout 300 python tools/kclock.py > gpurun_out/r03_final_inkernel_clock.txt 2>&1
python tools/kbench.py > gpurun_out/r03_final_kbench.txt 2>&1
python tools/slabstat.py > gpurun_out/r03_final_slabstat.txt 2>&1
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  timeout 600 python bench.py --workload $w --profile-ops --no-cpu-baseline --no-north-star-3ch > gpurun_out/r03_final_$w.json 2> gpurun_out/r03_final_$w.err
done
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-north-star-3ch > gpurun_out/r03_final_forcedist.json 2>/dev/null
ls -la gpurun_out | tail -5
