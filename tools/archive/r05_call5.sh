#!/bin/bash
# round 5, GPU call 5: whole suite, A/B tool (B = 8 split rotation, B = 16 split), kernel trace of the default bench, the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_c5_gpu_tests_tail.txt
cat gpurun_out/r05_c5_gpu_tests_tail.txt
python tools/r05_scan_ab.py 2>&1 | tail -1
bash tools/prof_bench.sh 2>&1 | tail -24
cp gpurun_out/prof_bench/*kernel_stats.csv gpurun_out/r05_c5_bench_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r05_c5_bench_kernel_stats.csv
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_c5_bench_default_line.json; cut -c1-1500 gpurun_out/r05_c5_bench_default_line.json
