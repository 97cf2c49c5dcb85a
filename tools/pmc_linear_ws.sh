#!/bin/bash
# PMC comparison on the in_proj shape (M = 65536, K = 640, N = 2560): linear_ws_kernel (weights stationary) against the tiled linear4w kernel
# as two half-width launches — L2 requests / hit rate, HBM bytes, MFMA busy, LDS activity (separate passes, counters only).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/lin_ws_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zigma_amd.linear import linear
x = torch.randn(65536, 640, device="cuda", dtype=torch.bfloat16); w = (torch.randn(2560, 640, device="cuda") * 0.04).bfloat16()
o = torch.empty(65536, 2560, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    linear(x, w, out=o, weight_stationary=True)
    linear(x, w[:1280], out=o[:, :1280]); linear(x, w[1280:], out=o[:, 1280:])
torch.cuda.synchronize()
PY
PASSES=("TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"
        "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM")
for pass in "${PASSES[@]}"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 120 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcws_$tag -o pmc -- python /tmp/lin_ws_one.py > $R/gpurun_out/pmcws_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcws_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "linear_ws" if "linear_ws" in n else "linear4w_half" if "linear4w" in n else None
        if key: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    out[k] = dict(launches=len(c["GRBM_GUI_ACTIVE"]), cycles=cyc, mfma_busy=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
                  wave_wait=m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                  l2_req=m.get("TCC_REQ_sum"), l2_hit_rate=m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
                  fetch_MB_x2=m.get("FETCH_SIZE", 0) * 2 / 1024, write_MB=m.get("WRITE_SIZE", 0) / 1024, lds_insts=m.get("SQ_INSTS_LDS"),
                  lds_idx_active=m.get("SQ_LDS_IDX_ACTIVE"), lds_bank_conflict=m.get("SQ_LDS_BANK_CONFLICT"), vmem_insts=m.get("SQ_INSTS_VMEM"), valu_insts=m.get("SQ_INSTS_VALU"))
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_linear_ws.json", "w"), indent=1)
PY
