# same-box A/B inside the forward: to_q (k = 640, n = 512) on the weight-stationary kernel (ZIGMA_TO_Q_WS=1) against the tiled kernel: a tie
# (17.50-17.57 vs 17.54-17.59 ms), like stand-alone (43-45 vs 41-43 us); the knob stays off
for rnd in 1 2 3; do
for v in 0 1; do
  echo -n "== ZIGMA_TO_Q_WS=$v: "
  ZIGMA_TO_Q_WS=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
done
