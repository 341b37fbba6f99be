cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { python bench.py --no-cpu-baseline --no-exact-fp32 --no-fit --no-north-star-3ch 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'])" "$1"; }
run base
ODIN_OVERLAP_WGRAD=small run overlap_small
ODIN_OVERLAP_WGRAD=all run overlap_all
ODIN_OVERLAP_WGRAD=small ODIN_EARLY_REDUCE=1 run small_early
ODIN_DEFER_WGRAD=1 run defer
run base
