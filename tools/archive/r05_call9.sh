#!/bin/bash
# round 5, GPU call 9: HEAD verification — whole suite, smoke, default bench line, kernel trace of config 3y with the six-resident scan
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_c9_gpu_tests_tail.txt
cat gpurun_out/r05_c9_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_c9_bench_default_line.json; cut -c1-400 gpurun_out/r05_c9_bench_default_line.json
bash tools/prof_configs.sh 3y 2>&1 | head -12
