#!/bin/bash
# round-2 evidence run on the GPU box: tests, default bench, rocprofv3 kernel trace, PMC passes, stamps,
# microbenchmarks, the other workloads, the RCCL path at world size 1.  Everything lands in gpurun_out/.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r02_final_gpu_tests.txt
python bench.py --profile-ops > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err
./tools/profile.sh r02_final_prof > /dev/null 2>&1
./tools/pmc.sh r02 > /dev/null 2>&1
python tools/pmc_traffic.py > /dev/null 2> gpurun_out/r02_final_pmc_traffic.err
python bench.py --no-cpu-baseline > gpurun_out/r02_final_bench2.json 2>/dev/null   # picks up the fresh traffic file
timeout 200 python tools/stamps_t.py > gpurun_out/r02_final_stamps_tconv_planes.txt 2>&1
timeout 200 python tools/stamps_dg.py > gpurun_out/r02_final_stamps_fconv_ring.txt 2>&1
timeout 200 python tools/wp_dbg.py > gpurun_out/r02_final_wgrad_planes_dbg.txt 2>&1
ODIN_TP_DBG=x DBGS="0 1 2 4 7" ./tools/tp_dbg.sh > gpurun_out/r02_final_tconv_planes_dbg.txt 2>&1
./tools/micro/run_all.sh
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r02_final_$w.json 2>/dev/null
done
timeout 300 python bench.py --gpus 1 --force-dist --no-cpu-baseline > gpurun_out/r02_final_forcedist.json 2>/dev/null
timeout 300 python bench.py --gpus 2 --dry-run > gpurun_out/r02_final_launcher_dryrun.txt 2>&1
ls -la gpurun_out | tail -5
