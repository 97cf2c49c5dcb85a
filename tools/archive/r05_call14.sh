#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
c5() { env "$@" python tools/run_configs.py --only 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(label='$LABEL', eager_ms=round(d['ms_per_forward'],3), hipgraph_ms=round(d['ms_per_forward_hipgraph'],3))))" | tee -a gpurun_out/r05_c14_config5_ab.jsonl; }
rm -f gpurun_out/r05_c14_config5_ab.jsonl
for rep in 1 2; do
LABEL=default c5 X=1
LABEL=linear_sm_off c5 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_TOKENS=False,model_zigma.TO_Q_FEW_TOKENS=False
LABEL=library_out_proj c5 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_TOKENS=False,mamba_simple.OUT_PROJ_WS_MAX_TOKENS=0
done
bash tools/prof_configs.sh 5 2>&1 | grep -v "^   void at\|^   __amd" | head -12
