#!/bin/bash
# round-6 evidence run on the GPU box: tests, default bench with per-op table, rocprofv3 kernel trace + timeline,
# FETCH / WRITE / SQ PMC passes -> measured HBM traffic of the priced kernels and of the whole step, plane-kernel SQ
# counters, the ELBO sweep, slab statistics, range-word fallbacks, the other workloads with timelines, the RCCL path at
# world size 1, the neck's in-kernel phase stamps, the same-call A/Bs of the round.  Everything lands in gpurun_out/;
# tools/r06_collect.sh copies what is to be judged into profiles/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
set -o pipefail
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r06_final_gpu_tests.txt
python bench.py --profile-ops > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
grep "^#" gpurun_out/r06_final_bench.err > gpurun_out/r06_final_per_op.txt
./tools/profile.sh r06_final_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
./tools/pmc.sh r06 --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
python tools/pmc_traffic.py r06 gpurun_out/r06_final_prof_timeline.txt > /dev/null 2> gpurun_out/r06_final_pmc_traffic.err || echo "PMC TRAFFIC TABLE STALE" >> gpurun_out/r06_final_pmc_traffic.err
cp profiles/r06_pmc_traffic.json gpurun_out/r06_pmc_traffic.json 2>/dev/null
python bench.py --no-cpu-baseline > gpurun_out/r06_final_bench2.json 2>/dev/null   # picks up the fresh traffic file
./tools/kpmc.sh r06f > /dev/null 2>&1
python tools/kbench.py > gpurun_out/r06_final_kbench.txt 2>&1
python tools/elbo_sweep6.py > gpurun_out/r06_final_elbo_sweep.txt 2>&1
python tools/range_fallbacks.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final_range_fallbacks.txt
python tools/slabstat.py > gpurun_out/r06_final_slabstat.txt 2>&1
python tools/stamps_neck.py > gpurun_out/r06_final_stamps_neck.txt 2>&1
python tools/stamps_neck.py shapes3d_vae_b256 >> gpurun_out/r06_final_stamps_neck.txt 2>&1
python tools/thinbench.py > gpurun_out/r06_final_thinbench.txt 2>&1
python tools/blkbench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final_blkbench.txt
python tools/densebench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_final_densebench.txt
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  timeout 600 python bench.py --workload $w --profile-ops --no-cpu-baseline --no-north-star-3ch > gpurun_out/r06_final_$w.json 2> gpurun_out/r06_final_$w.err
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512 mnist_conv_b128; do
  out=gpurun_out/r06_final_tl_$w; mkdir -p $out
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 --workload $w --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out.log 2>&1)
  per=1; [ $w = factorvae_shapes3d_b256 ] && per=2
  python tools/timeline.py $out 20 $per > gpurun_out/r06_final_tl_${w}_timeline.txt 2>&1
  rm -rf $out
done
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-north-star-3ch --no-fit --no-exact-fp32 > gpurun_out/r06_final_forcedist.json 2>/dev/null
# ---- same-call A/Bs ----
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ns=d.get('north_star_3ch') or {}; print(d['ms_per_step'], ns.get('ms_per_step'))"; }
{
  echo "# ms per step (second column: Shapes3D beta-VAE of the same line), alternating, one gpurun call"
  for i in 1 2 3; do
    echo "dsprites default                 $(ab)"
    echo "dsprites neck=False              $(ab --engine-opt neck=False --no-north-star-3ch)"
    echo "dsprites neck_bwd=False          $(ab --engine-opt neck_bwd=False --no-north-star-3ch)"
  done
  for i in 1 2; do
    echo "shapes3d default                 $(ab --workload shapes3d_vae_b256 --no-north-star-3ch)"
    echo "shapes3d neck=False              $(ab --workload shapes3d_vae_b256 --no-north-star-3ch --engine-opt neck=False)"
    echo "shapes3d neck_bwd=True           $(ab --workload shapes3d_vae_b256 --no-north-star-3ch --engine-opt neck_bwd=True)"
    echo "factorvae default                $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch)"
    echo "factorvae neck=False             $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch --engine-opt neck=False)"
  done
  for i in 1 2; do
    echo "speech default                   $(ab --workload speech_vae_b256 --no-north-star-3ch)"
    echo "speech --no-blk (igemm_h)        $(ab --workload speech_vae_b256 --no-north-star-3ch --no-blk)"
  done
  for i in 1 2; do
    echo "speech default                   $(ab --workload speech_vae_b256 --no-north-star-3ch)"
    echo "speech --no-mel-r16              $(ab --workload speech_vae_b256 --no-north-star-3ch --no-mel-r16)"
    echo "factorvae default                $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch)"
    echo "factorvae --no-fused-disc        $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch --no-fused-disc)"
    echo "factorvae --no-dense-hw          $(ab --workload factorvae_shapes3d_b256 --no-north-star-3ch --no-dense-hw)"
    echo "celeba default                   $(ab --workload celeba_betatcvae_b512 --no-north-star-3ch)"
    echo "celeba --no-dense-hw             $(ab --workload celeba_betatcvae_b512 --no-north-star-3ch --no-dense-hw)"
  done
  echo "dsprites early_reduce=True       $(ab --engine-opt early_reduce=True --no-north-star-3ch)"
  echo "dsprites hyper_ring=False        $(ab --engine-opt hyper_ring=False --no-north-star-3ch)"
  echo "dsprites act_words=False         $(ab --engine-opt act_words=False --no-north-star-3ch)"
  echo "dsprites fuse_norm=False         $(ab --engine-opt fuse_norm=False --no-north-star-3ch)"
  echo "dsprites default                 $(ab --no-north-star-3ch)"
} > gpurun_out/r06_final_ab.txt 2>&1
ls gpurun_out | grep r06_final | wc -l
