#!/bin/bash
# MFMA utilisation of every projection kernel inside the real forward (library and own), from rocprofv3 PMC passes over two forwards
# of the headline model (counters in their own runs, no trace domains).  Prints per kernel: launches, mean duration in shader cycles,
# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs), achieved TFLOP/s is in profiles/*linear_probe*.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
  tag=pj_$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/tools/fwd_one.py > $R/gpurun_out/pmc_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_pj_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "Cijk" in n or "linear4w" in n or "linear_tn" in n or "dt_proj" in n or "conv_x_proj" in n or "cross_attn" in n:
            key = n.split("(")[0][:60] if "Cijk" not in n else n[:70]
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    g = c.get("GRBM_GUI_ACTIVE"); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if not g or not mf: continue
    cyc = sum(g) / len(g) / 8
    busy = sum(mf) / len(mf) / (cyc * 1024)
    wait = sum(c["SQ_WAIT_ANY"]) / sum(c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else float("nan")
    print(f"{k:72s} n={len(g):4d} cycles={cyc:9.0f} mfma_busy={busy:5.3f} wave_wait={wait:5.3f}")
PY
