#!/bin/bash
# PMC counters for the dt_proj kernel alone (separate passes; no trace domains combined with --pmc).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmcdt_$tag -o pmc -- python $R/tools/dt_probe.py > $R/gpurun_out/pmcdt_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcdt_*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "dt_proj" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
tail -3 gpurun_out/pmcdt_SQ_VALU_MFMA_BUSY_CYCLES.log
