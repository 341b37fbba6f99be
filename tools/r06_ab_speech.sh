ab() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --workload speech_vae_b256 --no-north-star-3ch 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2 3; do
  echo "new  $(ab)"
  echo "prev $(ODIN_HIP_LIB=tools/diag/libodin_prev.so ab)"
done
