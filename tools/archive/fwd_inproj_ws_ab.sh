# same-box A/B inside the forward (bench.py, 20 steps): in_proj on the weight-stationary kernel (one launch) vs as two half-width launches of
# the tiled kernel vs on the library
for v in "1:1" "0:1" "0:0" "1:1" "0:1" "0:0"; do
  ws=${v%%:*}; sp=${v##*:}
  echo "== ZIGMA_IN_PROJ_WS=$ws ZIGMA_IN_PROJ_SPLIT=$sp"
  ZIGMA_IN_PROJ_WS=$ws ZIGMA_IN_PROJ_SPLIT=$sp python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
