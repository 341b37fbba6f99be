#!/bin/bash
# copies the outputs of tools/r06_final.sh (gpurun_out/, scratch) to their committed names under profiles/
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/r06_final_bench.json $p/r06_bench_default.json
cp $g/r06_final_bench2.json $p/r06_bench_default_fresh_traffic.json
cp $g/r06_final_per_op.txt $p/r06_bench_per_op.txt
cp $g/r06_final_gpu_tests.txt $p/r06_gpu_tests.txt
cp $g/r06_final_prof_kernel_stats.csv $p/r06_kernel_stats.csv
cp $g/r06_final_prof_summary.txt $p/r06_kernel_stats_summary.txt
cp $g/r06_final_prof_timeline.txt $p/r06_step_timeline.txt
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES; do cp $g/r06_pmc_$c.txt $p/r06_pmc_$c.txt; done
cat $g/r06f_kpmc_1.txt $g/r06f_kpmc_2.txt $g/r06f_kpmc_3.txt $g/r06f_kpmc_4.txt > $p/r06_kpmc_planes_final.txt
cp $g/r06_final_kbench.txt $p/r06_kbench.txt
cp $g/r06_final_elbo_sweep.txt $p/r06_elbo_sweep.txt
cp $g/r06_final_range_fallbacks.txt $p/r06_range_fallbacks.txt
cp $g/r06_final_slabstat.txt $p/r06_slabstat.txt
cp $g/r06_final_stamps_neck.txt $p/r06_stamps_neck.txt
cp $g/r06_final_thinbench.txt $p/r06_thinbench.txt
cp $g/r06_final_blkbench.txt $p/r06_blkbench.txt
cp $g/r06_final_densebench.txt $p/r06_dense_hw_bench.txt
cp $g/r06_final_ab.txt $p/r06_ab_same_call.txt
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  cp $g/r06_final_$w.json $p/r06_bench_$w.json
  grep "^#" $g/r06_final_$w.err > $p/r06_bench_${w}_per_op.txt
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512 mnist_conv_b128; do
  cp $g/r06_final_tl_${w}_timeline.txt $p/r06_step_timeline_$w.txt
done
cp $g/r06_final_forcedist.json $p/r06_bench_force_dist_rccl.json
cp $g/r06_pmc_traffic.json $p/r06_pmc_traffic.json
