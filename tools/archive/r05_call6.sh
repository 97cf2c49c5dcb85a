#!/bin/bash
# round 5, GPU call 6: the state to be recorded — whole suite, default bench line, kernel traces (bench, configs 3y / 4 / 5, B = 8 / 16), configs, batch sweep
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_c6_gpu_tests_tail.txt
cat gpurun_out/r05_c6_gpu_tests_tail.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r05_c6_bench_default_line.json; cut -c1-700 gpurun_out/r05_c6_bench_default_line.json
python bench.py --no-cpu-baseline --scan-events-every 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('events every launch:', d['ms_per_step'], d['roofline']['launch_us'])"
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('events every 7th   :', d['ms_per_step'], d['roofline']['launch_us'], d['roofline']['launches'])"
bash tools/prof_bench.sh 2>&1 | tail -12
cp $(find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/r05_c6_bench_kernel_stats.csv
bash tools/prof_configs.sh 3y 4 5 b8 b16 2>&1 | tee gpurun_out/r05_c6_prof_configs.txt | grep -v "^   void\|^   __amd\|^   zigma" 
for c in 3 3y 4 5; do python tools/run_configs.py --only $c 2>/dev/null | cut -c1-700; done | tee gpurun_out/r05_c6_configs.jsonl
rm -f gpurun_out/r05_c6_batch_sweep.jsonl
for b in 8 16 32 128; do python bench.py --no-cpu-baseline --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=$b, ms_per_step=d['ms_per_step'], tokens_per_s=d['value'], check=d['check']['rel_err_vs_unfused'])))" | tee -a gpurun_out/r05_c6_batch_sweep.jsonl; done
