cd ${GRAFT_REPO_ROOT:-$(pwd)}
one() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for w in celeba_betatcvae_b512 factorvae_shapes3d_b256 shapes3d_vae_b256; do
  for i in 1 2; do echo "$w 8-wave $(one --workload $w)  pc $(ODIN_BP_PC=1 one --workload $w)"; done
done
