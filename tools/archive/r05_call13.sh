#!/bin/bash
# round 5, GPU call 13: config 5 and B = 8 with the few-token kernel on / off / library routing (one box), B = 8 trace, whole suite
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
c5() { env "$@" python tools/run_configs.py --only 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(label='$LABEL', eager_ms=round(d['ms_per_forward'],3), hipgraph_ms=round(d['ms_per_forward_hipgraph'],3))))" | tee -a gpurun_out/r05_c13_config5_ab.jsonl; }
rm -f gpurun_out/r05_c13_config5_ab.jsonl
for rep in 1 2; do
LABEL=default c5 X=1
LABEL=linear_sm_off c5 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_TOKENS=False,model_zigma.TO_Q_FEW_TOKENS=False
LABEL=library_out_proj c5 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_TOKENS=False,mamba_simple.OUT_PROJ_WS_MAX_TOKENS=0
done
bash tools/prof_configs.sh b8 5 2>&1 | grep -v "^   void at\|^   __amd"
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05_c13_gpu_tests_tail.txt; cat gpurun_out/r05_c13_gpu_tests_tail.txt
