#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench run -> gpurun_out/prof_bench/ (copy the stats csv into profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-check --steps 10 --warmup 3 > $R/gpurun_out/prof_bench_line.json 2> $R/gpurun_out/prof_bench.err
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_bench/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f} pct={float(r["TotalDurationNs"])/tot*100:5.1f}')
PY
