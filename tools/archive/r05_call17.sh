#!/bin/bash
# round 5, GPU call 17: the few-token kernel's fused gated add against its plain product + the add in the next norm kernel (B = 8, config 5; hipGraph and eager)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
rm -f gpurun_out/r05_c17_few_token_fuse_ab.jsonl
for rep in 1 2; do
for v in True False; do
  ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_FUSE=$v python tools/latency_probe.py 2>/dev/null | grep '"batch": 8' | sed "s/^{/{\"out_proj_few_fuse\": \"$v\", /" | tee -a gpurun_out/r05_c17_few_token_fuse_ab.jsonl
  ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FEW_FUSE=$v python tools/run_configs.py --only 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict(out_proj_few_fuse='$v', config=5, eager_ms=round(d['ms_per_forward'],3), hipgraph_ms=round(d['ms_per_forward_hipgraph'],3))))" | tee -a gpurun_out/r05_c17_few_token_fuse_ab.jsonl
done; done
