# same-box A/B of the in_proj variants inside the forward (bench.py, 20 steps): library GEMM / two half-width launches of the own
# kernel / x half inside the conv + x_proj kernel (zigma_in_conv_x_proj_fwd)
for v in "0:0" "1:0" "0:1" "0:0" "1:0" "0:1"; do
  sp=${v%%:*}; ic=${v##*:}
  echo "== ZIGMA_IN_PROJ_SPLIT=$sp ZIGMA_IN_CONV_X_PROJ=$ic"
  ZIGMA_IN_PROJ_SPLIT=$sp ZIGMA_IN_CONV_X_PROJ=$ic python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check'])"
done
