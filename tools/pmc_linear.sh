#!/bin/bash
# L2 hit rate / HBM bytes of zigma_linear_fwd at the in_proj shape (separate PMC passes).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lin_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zigma_amd.linear import linear
x = torch.randn(65536, 640, device="cuda", dtype=torch.bfloat16); w = (torch.randn(2560, 640, device="cuda") * 0.04).bfloat16()
for _ in range(10): linear(x, w, None, 1280)
torch.cuda.synchronize()
PY
PASSES=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
        "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM")
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcl_$tag -o pmc -- python /tmp/lin_one.py > $R/gpurun_out/pmcl_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcl_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "linear_tn" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
