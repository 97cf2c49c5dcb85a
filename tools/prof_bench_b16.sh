R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -o bench -- python $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 --batch 16 > $R/gpurun_out/prof_b16_line.json 2> $R/gpurun_out/prof_b16.err
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_b16/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total busy ms per forward ~", tot / 14 / 1e6)
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f} pct={float(r["TotalDurationNs"])/tot*100:5.1f}')
PY
tail -c 300 $R/gpurun_out/prof_b16_line.json | head -c 200
