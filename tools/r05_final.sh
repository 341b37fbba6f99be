#!/bin/bash
# round-5 evidence run on the GPU box: tests, default bench with per-op table, rocprofv3 kernel trace + timeline,
# FETCH / WRITE / SQ PMC passes -> measured HBM traffic of the dominant kernels AND of the whole step (fails when a priced
# kernel no longer exists), plane-kernel SQ counters, the stream probes behind the ELBO kernel's ceiling, the other
# workloads, the RCCL path at world size 1, the same-call A/Bs of the round (hyper-parameter ring, activation range
# words, the engine's overlap options).  Everything lands in gpurun_out/; copy what is to be judged into profiles/
# (tools/r05_collect.sh).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
set -o pipefail
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -n 2 2>&1 | tail -3 > gpurun_out/r05_final_gpu_tests.txt
python bench.py --profile-ops > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
grep "^#" gpurun_out/r05_final_bench.err > gpurun_out/r05_final_per_op.txt
./tools/profile.sh r05_final_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
./tools/pmc.sh r05 --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
python tools/pmc_traffic.py r05 gpurun_out/r05_final_prof_timeline.txt > /dev/null 2> gpurun_out/r05_final_pmc_traffic.err || echo "PMC TRAFFIC TABLE STALE" >> gpurun_out/r05_final_pmc_traffic.err
cp profiles/r05_pmc_traffic.json gpurun_out/r05_pmc_traffic.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/r05_final_bench2.json 2>/dev/null   # picks up the fresh traffic file
./tools/kpmc.sh r05f > /dev/null 2>&1
python tools/kbench.py > gpurun_out/r05_final_kbench.txt 2>&1
python tools/elbo_ceiling.py > gpurun_out/r05_final_elbo_stream_sweep.txt 2>&1
python tools/range_fallbacks.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_final_range_fallbacks.txt
python tools/slabstat.py > gpurun_out/r05_final_slabstat.txt 2>&1
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  timeout 600 python bench.py --workload $w --profile-ops --no-cpu-baseline --no-north-star-3ch > gpurun_out/r05_final_$w.json 2> gpurun_out/r05_final_$w.err
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512; do
  ./tools/profile.sh r05_final_tl_$w --workload $w --no-north-star-3ch --no-fit > /dev/null 2>&1
done
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 > gpurun_out/r05_final_forcedist.json 2>/dev/null
# ---- same-call A/Bs ----
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
{
  echo "# dsprites_betavae_b256 (ms per step, Shapes3D ms per step), alternating, one gpurun call"
  for i in 1 2 3; do
    echo "default              $(ab)"
    echo "ODIN_HYPER_RING=0    $(ODIN_HYPER_RING=0 ab)"
    echo "ODIN_ACT_WORDS=0     $(ODIN_ACT_WORDS=0 ab)"
  done
  echo "ODIN_OVERLAP_WGRAD=small                    $(ODIN_OVERLAP_WGRAD=small ab --no-north-star-3ch)"
  echo "ODIN_OVERLAP_WGRAD=all                      $(ODIN_OVERLAP_WGRAD=all ab --no-north-star-3ch)"
  echo "ODIN_OVERLAP_WGRAD=small ODIN_EARLY_REDUCE=1 $(ODIN_OVERLAP_WGRAD=small ODIN_EARLY_REDUCE=1 ab --no-north-star-3ch)"
  echo "ODIN_DEFER_WGRAD=1                          $(ODIN_DEFER_WGRAD=1 ab --no-north-star-3ch)"
  echo "default                                     $(ab --no-north-star-3ch)"
} > gpurun_out/r05_final_ab.txt 2>&1
ls -la gpurun_out | tail -5
