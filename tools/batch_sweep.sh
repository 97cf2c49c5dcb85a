#!/bin/bash
# bench.py at per-GPU batches 8 / 16 / 32 / 64 / 128 (with its reference check where the batch allows) + the serving-latency probe (eager / hipGraph, B = 1 .. 16)
# -> gpurun_out/<prefix>_batch_sweep.jsonl, <prefix>_serving_latency.jsonl.   usage: tools/batch_sweep.sh [prefix]
R=${GRAFT_REPO_ROOT:-$PWD}; P=${1:-sweep}
rm -f $R/gpurun_out/${P}_batch_sweep.jsonl
for b in 8 16 32 64 128; do
  python $R/bench.py --no-cpu-baseline --batch $b --steps 20 2>/dev/null | python -c "import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=$b, ms_per_step=d['ms_per_step'], tokens_per_s=d['value'], scan_us=d['roofline']['launch_us'], rel_err_vs_unfused=d['check']['rel_err_vs_unfused'], rel_err_vs_reference_fp32=d['check'].get('rel_err_vs_reference_fp32'), box=d.get('box_calib', {}).get('copy_1GiB_GBps'))))" | tee -a $R/gpurun_out/${P}_batch_sweep.jsonl
done
python $R/tools/latency_probe.py 2>/dev/null | tee $R/gpurun_out/${P}_serving_latency.jsonl
