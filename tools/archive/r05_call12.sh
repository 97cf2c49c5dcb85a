#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -x -q -m gpu -k "linear_sm or bench_block_path or no_text or linear_ws_kernel" 2>&1 | tail -4
python tools/latency_probe.py 2>/dev/null | tail -2
ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_TOKENS=False,model_zigma.TO_Q_FEW_TOKENS=False python tools/latency_probe.py 2>/dev/null | tail -2
python tools/run_configs.py --only 5 2>/dev/null | cut -c100-420
for b in 8 16; do python bench.py --no-cpu-baseline --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=$b, ms_per_step=d['ms_per_step'], check=d['check']['rel_err_vs_unfused'])))"; done
