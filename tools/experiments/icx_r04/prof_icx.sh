#!/bin/bash
# per-kernel GPU durations of in_conv_x_proj (all timing-probe variants) from rocprofv3 --kernel-trace --stats.  Driver: tools/icx_one.py
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for fl in ${FLAGS_LIST:-0 2 4 8 12 16 30 32 62}; do
  FLAGS=$fl N=12 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_icx_$fl -o p -- python $R/tools/icx_one.py > $R/gpurun_out/prof_icx_$fl.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, re
for d in sorted(glob.glob("gpurun_out/prof_icx_*/"), key=lambda s: int(re.findall(r"_(\d+)/", s)[0])):
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "in_conv_x_proj" in r["Name"] or "in_halo" in r["Name"]:
                print(d.split("/")[-2], r["Name"][:60], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3, "min_us", float(r["MinNs"]) / 1e3)
PY
