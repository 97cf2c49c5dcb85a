#!/bin/bash
# round 5, GPU call 16: which of the three few-token routes costs the B = 16 forward (16 384 tokens: two rounds of tiles)?
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
fwd() {
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --batch 16 --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(label='$label', ms_per_step=round(d['ms_per_step'],3))))" | tee -a gpurun_out/r05_c16_b16_few_token_routes.jsonl
}
rm -f gpurun_out/r05_c16_b16_few_token_routes.jsonl
for rep in 1 2; do
fwd default X=1
fwd to_q_and_to_out_sm ZIGMA_KNOBS=model_zigma.TO_Q_FEW_MAX_TOKENS=16384
fwd out_proj_sm ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_MAX_TOKENS=16384
fwd all_sm ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_MAX_TOKENS=16384,model_zigma.TO_Q_FEW_MAX_TOKENS=16384
done
