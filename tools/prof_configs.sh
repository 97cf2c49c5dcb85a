#!/bin/bash
# rocprofv3 kernel-trace summaries of BASELINE configs 3 (the shipped yaml model, E=768 depth 24), 4, 5 and of the serving-size batches
# (bench.py --batch 8 / 16): one kernel_stats csv each -> gpurun_out/prof_<tag>/ ; prints the top kernels and the number of library GEMM rows.
# usage: tools/prof_configs.sh [tags...]   (default: 3y v2 4 5 b8 b16)
R=${GRAFT_REPO_ROOT:-$PWD}
TAGS=${@:-3y v2 4 5 b8 b16}
cd /tmp && export TMPDIR=/tmp
for tag in $TAGS; do
  rm -rf $R/gpurun_out/prof_$tag
  case $tag in
    b*) CMD="python $R/bench.py --no-cpu-baseline --no-check --batch ${tag#b} --steps 10 --warmup 3" ;;
    *)  CMD="python $R/tools/run_configs.py --only $tag" ;;
  esac
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o t -- $CMD > $R/gpurun_out/prof_${tag}_line.json 2> $R/gpurun_out/prof_$tag.err
  tail -1 $R/gpurun_out/prof_${tag}_line.json | cut -c1-400
  python - $R $tag <<'PY'
import csv, glob, sys, shutil
R, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{R}/gpurun_out/prof_{tag}/**/*kernel_stats.csv", recursive=True)
if not f:
    print(tag, "no kernel_stats.csv"); sys.exit(0)
shutil.copy(f[0], f"{R}/gpurun_out/r06_cfg_{tag}_kernel_stats.csv")
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lib = [r for r in rows if "Cijk_" in r["Name"]]       # (hipBLASLt kernels: "Cijk_..." and "Custom_Cijk_...")
print(f"== {tag}: {len(rows)} kernels, library GEMM rows (Cijk_*): {len(lib)}, their share {sum(float(r['TotalDurationNs']) for r in lib) / tot * 100:.1f} %")
for r in rows[:12]:
    print(f'   {r["Name"][:64]:64s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f} pct={float(r["TotalDurationNs"])/tot*100:5.1f}')
for r in lib:
    print(f'   LIB {r["Name"][:90]:90s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
done
