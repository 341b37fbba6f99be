#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r06p10_tests.txt
python bench.py --no-cpu-baseline --profile-ops > gpurun_out/r06p10_bench.json 2> gpurun_out/r06p10_bench.err
./tools/profile.sh r06p10_prof --no-north-star-3ch --no-fit --no-exact-fp32 > /dev/null 2>&1
cat gpurun_out/r06p10_tests.txt; cut -c1-300 gpurun_out/r06p10_bench.json; grep "^#" gpurun_out/r06p10_bench.err | cut -c1-120; cat gpurun_out/r06p10_prof_timeline.txt | cut -c1-110
