#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
fwd() {
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --batch 16 --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(label='$label', ms_per_step=round(d['ms_per_step'],3))))" | tee -a gpurun_out/r05_c18_b16_unfused_sm.jsonl
}
rm -f gpurun_out/r05_c18_b16_unfused_sm.jsonl
for rep in 1 2; do
fwd default X=1
fwd out_proj_sm_unfused_16384 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_MAX_TOKENS=16384
fwd out_proj_and_to_q_sm_16384 ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_MAX_TOKENS=16384,model_zigma.TO_Q_FEW_MAX_TOKENS=16384
done
python -m pytest tests -x -q -m gpu -k "bench_block_path or no_text or linear_sm" 2>&1 | tail -3
