# same-box A/B of the forward (bench.py, 20 steps): dt_proj + softplus inside the scan kernel vs the separate dt_proj kernel
for v in 0 1 0 1; do
  echo "== ZIGMA_DT_IN_SCAN=$v"
  ZIGMA_DT_IN_SCAN=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check'])"
done
