#!/bin/bash
# round 5, GPU call 8: the six-resident-workgroups form of the scan (E = 768 models): tests, A/B, config 3y
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan or small_batch or mamba_inner" 2>&1 | tail -5
python tools/r05_scan_ab.py 2>&1 | tail -1
python tools/run_configs.py --only 3y 2>/dev/null | cut -c1-400
python tools/run_configs.py --only 5 2>/dev/null | cut -c1-500
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
