#!/bin/bash
# One GPU session that produces every measurement the round-2 docs quote (outputs under gpurun_out/r02/).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -4 > $O/gpu_tests.txt
timeout 200 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 120 python tools/scan_ab.py > $O/scan_ab.json 2>/dev/null
timeout 120 python tools/conv_xproj_ab.py 2>/dev/null | grep "^{" > $O/conv_xproj_ab.json
timeout 200 python tools/linear_probe.py 2>/dev/null | grep "^{" > $O/linear_probe.jsonl
timeout 250 python tools/run_configs.py 2>/dev/null > $O/configs_3_4_5.jsonl
bash tools/prof_bench.sh > $O/prof_bench_top.txt 2>&1
cp gpurun_out/prof_bench/bench_kernel_stats.csv $O/bench_kernel_stats.csv
cp gpurun_out/prof_bench_line.json $O/bench_profiled_line.json
bash tools/pmc_scan.sh > $O/pmc_scan.txt 2>&1
tail -3 $O/gpu_tests.txt; cat $O/bench_default_line.json | cut -c1-400; tail -2 $O/bench_default.err
