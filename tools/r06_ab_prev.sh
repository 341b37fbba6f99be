# same-call A/B of two builds of the library: the current one against tools/diag/libodin_prev.so (copied by hand before a change)
ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for w in ${AB_WORKLOADS:-dsprites_betavae_b256 celeba_betatcvae_b512 shapes3d_vae_b256}; do
  for i in 1 2 3; do
    echo "$w new  $(ab --workload $w)"
    echo "$w prev $(ODIN_HIP_LIB=tools/diag/libodin_prev.so ab --workload $w)"
  done
done
