#!/bin/bash
# round-4 evidence run on the GPU box: tests, default bench with per-op table, rocprofv3 kernel trace + timeline,
# FETCH / WRITE / SQ PMC passes -> measured HBM traffic of the dominant kernels (fails when a priced kernel no
# longer exists), plane-kernel SQ counters, in-kernel stamps (diagnostics build), the stream probes behind the ELBO
# kernel's ceiling, the other workloads, the RCCL path at world size 1.  Everything lands in gpurun_out/; copy what is
# to be judged into profiles/ (tools/r04_collect.sh).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
set -o pipefail
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r04_final_gpu_tests.txt
python bench.py --profile-ops > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
grep "^#" gpurun_out/r04_final_bench.err > gpurun_out/r04_final_per_op.txt
./tools/profile.sh r04_final_prof --no-north-star-3ch --no-fit > /dev/null 2>&1
./tools/pmc.sh r04 --no-north-star-3ch --no-fit > /dev/null 2>&1
python tools/pmc_traffic.py r04 > /dev/null 2> gpurun_out/r04_final_pmc_traffic.err || echo "PMC TRAFFIC TABLE STALE" >> gpurun_out/r04_final_pmc_traffic.err
cp profiles/r04_pmc_traffic.json gpurun_out/r04_pmc_traffic.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/r04_final_bench2.json 2>/dev/null   # picks up the fresh traffic file
./tools/kpmc.sh r04f > /dev/null 2>&1
timeout 200 python tools/stamps_fp.py > gpurun_out/r04_final_stamps_fconv_planes.txt 2>&1
timeout 200 python tools/stamps_tail.py > gpurun_out/r04_final_stamps_tail.txt 2>&1
python tools/kbench.py > gpurun_out/r04_final_kbench.txt 2>&1
python tools/elbo_ceiling.py > gpurun_out/r04_final_elbo_stream_sweep.txt 2>&1
python tools/range_fallbacks.py > gpurun_out/r04_final_range_fallbacks.txt 2>&1
python tools/slabstat.py > gpurun_out/r04_final_slabstat.txt 2>&1
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  timeout 600 python bench.py --workload $w --profile-ops --no-cpu-baseline --no-north-star-3ch > gpurun_out/r04_final_$w.json 2> gpurun_out/r04_final_$w.err
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512; do
  ./tools/profile.sh r04_final_tl_$w --workload $w --no-north-star-3ch --no-fit > /dev/null 2>&1
done
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-north-star-3ch --no-fit > gpurun_out/r04_final_forcedist.json 2>/dev/null
ls -la gpurun_out | tail -5
