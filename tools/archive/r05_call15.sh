#!/bin/bash
# round 5, GPU call 15: HEAD — whole suite, smoke, default bench line (+ profiled kernel stats), batch sweep, serving latency
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05_c15_gpu_tests_tail.txt; cat gpurun_out/r05_c15_gpu_tests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_c15_bench_default_line.json; cut -c1-330 gpurun_out/r05_c15_bench_default_line.json
bash tools/prof_bench.sh 2>&1 | tail -22 | head -10
cp $(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r05_c15_bench_kernel_stats.csv
rm -f gpurun_out/r05_c15_batch_sweep.jsonl
for b in 8 16 32 128; do python bench.py --no-cpu-baseline --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=$b, ms_per_step=d['ms_per_step'], tokens_per_s=d['value'], check=d['check']['rel_err_vs_unfused'])))" | tee -a gpurun_out/r05_c15_batch_sweep.jsonl; done
python tools/latency_probe.py 2>/dev/null | tee gpurun_out/r05_c15_serving_latency.jsonl
