#!/bin/bash
# PMC counters for the scan backward kernel (separate passes; no trace domains combined with --pmc).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  B=64 timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcbwd_$tag -o pmc -- python $R/tools/bwd_probe.py > $R/gpurun_out/pmcbwd_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcbwd_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "scan_bwd_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
PY
