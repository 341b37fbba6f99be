#!/bin/bash
# copies the outputs of tools/r05_final.sh (gpurun_out/, scratch) to their committed names under profiles/
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/r05_final_bench.json $p/r05_bench_default.json
cp $g/r05_final_bench2.json $p/r05_bench_default_fresh_traffic.json
cp $g/r05_final_per_op.txt $p/r05_bench_per_op.txt
cp $g/r05_final_gpu_tests.txt $p/r05_gpu_tests.txt
cp $g/r05_final_prof_kernel_stats.csv $p/r05_kernel_stats.csv
cp $g/r05_final_prof_summary.txt $p/r05_kernel_stats_summary.txt
cp $g/r05_final_prof_timeline.txt $p/r05_step_timeline.txt
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES; do cp $g/r05_pmc_$c.txt $p/r05_pmc_$c.txt; done
cat $g/r05f_kpmc_1.txt $g/r05f_kpmc_2.txt $g/r05f_kpmc_3.txt $g/r05f_kpmc_4.txt > $p/r05_kpmc_planes_final.txt
cp $g/r05_final_kbench.txt $p/r05_kbench.txt
cp $g/r05_final_elbo_stream_sweep.txt $p/r05_elbo_stream_sweep.txt
cp $g/r05_final_range_fallbacks.txt $p/r05_range_fallbacks.txt
cp $g/r05_final_slabstat.txt $p/r05_slabstat.txt
cp $g/r05_final_ab.txt $p/r05_ab_same_call.txt
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 mnist_conv_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  cp $g/r05_final_$w.json $p/r05_bench_$w.json
  grep "^#" $g/r05_final_$w.err > $p/r05_bench_${w}_per_op.txt
done
for w in speech_vae_b256 factorvae_shapes3d_b256 celeba_betatcvae_b512; do
  cp $g/r05_final_tl_${w}_timeline.txt $p/r05_step_timeline_$w.txt
done
cp $g/r05_final_forcedist.json $p/r05_bench_force_dist_rccl.json
cp $g/r05_pmc_traffic.json $p/r05_pmc_traffic.json
