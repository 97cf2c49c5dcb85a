#!/bin/bash
# round 5, GPU call 3: routing policy (E=768 in_proj as one launch, to_q weight-stationary below 32 768 tokens, out_proj's add unfused below the
# 4-wave floor), split-K x_proj: tests, kernel traces of configs 3y / 4 / 5 / B=8 / B=16, configs line, PMC of the scan
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "x_proj or bench_block_path or linear_ws or scan_dt or no_text" 2>&1 | tail -6 > gpurun_out/r05_c3_tests.txt
cat gpurun_out/r05_c3_tests.txt
bash tools/prof_configs.sh 3y 4 5 b8 b16 2>&1 | tee gpurun_out/r05_c3_prof_configs.txt
python tools/run_configs.py 2>/dev/null > gpurun_out/r05_c3_configs.jsonl; cat gpurun_out/r05_c3_configs.jsonl | cut -c1-600
for b in 8 16 32; do python bench.py --no-cpu-baseline --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=$b, ms_per_step=d['ms_per_step'], tokens_per_s=d['value'], check=d['check']['rel_err_vs_unfused'])))" | tee -a gpurun_out/r05_c3_batch_sweep.jsonl; done
bash tools/pmc_scan.sh > gpurun_out/r05_c3_pmc_scan.txt 2>&1; tail -40 gpurun_out/r05_c3_pmc_scan.txt
