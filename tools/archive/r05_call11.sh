#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -x -q -m gpu -k "linear_sm or bench_block_path" 2>&1 | tail -3
python tools/r05_shapes_probe.py 2>&1 | grep "E640" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['us'])"
