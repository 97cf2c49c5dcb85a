#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
B=64 timeout 200 python $R/tools/train_probe.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_train
B=64 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o train -- python $R/tools/train_probe.py > $R/gpurun_out/prof_train.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms per step", tot / 7 / 1e6)
for r in rows[:26]:
    print(f'{r["Name"][:86]:86s} calls={int(r["Calls"])/7:7.1f} us/step={float(r["TotalDurationNs"])/7/1e3:9.1f} pct={float(r["TotalDurationNs"])/tot*100:5.1f}')
PY
