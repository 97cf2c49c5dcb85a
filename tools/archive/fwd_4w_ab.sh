# same-box A/B of the projection policy inside the forward (bench.py, 20 steps): 4-wave kernel vs 8-wave kernel for the epilogue
# projections, and in_proj on the 4-wave kernel vs the library
for v in "0:1024" "1:1024" "0:1024" "1:1024" "0:4096"; do
  w8=${v%%:*}; mx=${v##*:}
  echo "== ZIGMA_LINEAR_8W=$w8 ZIGMA_4W_MAX_N=$mx"
  ZIGMA_LINEAR_8W=$w8 ZIGMA_4W_MAX_N=$mx python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'])"
done
