#!/bin/bash
# copies the outputs of tools/r03_final.sh (gpurun_out/, scratch) to their committed names under profiles/
cd "$(dirname "$0")/.."
g=gpurun_out; p=profiles
cp $g/r03_final_bench.json $p/r03_bench_default.json
cp $g/r03_final_bench2.json $p/r03_bench_default_fresh_traffic.json
cp $g/r03_final_per_op.txt $p/r03_bench_per_op.txt
cp $g/r03_final_gpu_tests.txt $p/r03_gpu_tests.txt
cp $g/r03_final_prof_kernel_stats.csv $p/r03_kernel_stats.csv
cp $g/r03_final_prof_summary.txt $p/r03_kernel_stats_summary.txt
cp $g/r03_final_prof_timeline.txt $p/r03_step_timeline.txt
for c in FETCH_SIZE WRITE_SIZE SQ_WAVES; do cp $g/r03_pmc_$c.txt $p/r03_pmc_$c.txt; done
cat $g/r03f_kpmc_1.txt $g/r03f_kpmc_2.txt $g/r03f_kpmc_3.txt $g/r03f_kpmc_4.txt > $p/r03_kpmc_planes_final.txt
cp $g/r03_final_stamps_igemm.txt $p/r03_stamps_igemm.txt
cp $g/r03_final_stamps_fconv_planes.txt $p/r03_stamps_fconv_planes.txt
cp $g/r03_final_igbench.txt $p/r03_igbench.txt
cp $g/r03_final_igbench_tiled.txt $p/r03_igbench_tiled_paths.txt
cp $g/r03_final_enc0bench.txt $p/r03_enc0bench.txt
cp $g/r03_final_kbench.txt $p/r03_kbench.txt
cp $g/r03_final_slabstat.txt $p/r03_slabstat.txt
cp $g/r03_final_inkernel_clock.txt $p/r03_inkernel_clock.txt
for w in shapes3d_vae_b256 celeba_betatcvae_b512 mnist_dense_b128 factorvae_shapes3d_b256 speech_vae_b256; do
  cp $g/r03_final_$w.json $p/r03_bench_$w.json
  grep "^#" $g/r03_final_$w.err > $p/r03_bench_${w}_per_op.txt
done
cp $g/r03_final_forcedist.json $p/r03_bench_force_dist_rccl.json
python tools/pmc_traffic.py r03 > /dev/null
