#!/bin/bash
# round 5, GPU call 10: the few-token tiled kernel (linear_sm): tests, shapes probe, serving latency, B = 8 / 16 trace
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "linear_sm or bench_block_path or no_text" 2>&1 | tail -5
python tools/r05_shapes_probe.py 2>&1 | grep "out_proj\|to_q_E640 M=8192\|in_proj_E640 M=8192" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['us'])"
python tools/latency_probe.py 2>/dev/null | tail -2
python tools/run_configs.py --only 5 2>/dev/null | cut -c100-420
