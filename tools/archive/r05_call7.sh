#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "bench_block_path or no_text" 2>&1 | tail -4
python tools/r05_fuse_ab.py 2>&1 | tail -1
bash tools/prof_configs.sh b16 2>&1 | grep -v "^   void\|^   __amd\|^   zigma::"
python bench.py --no-cpu-baseline --batch 16 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(batch=16, ms_per_step=d['ms_per_step'])))"
