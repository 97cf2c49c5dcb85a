# same-box A/B inside the forward (bench.py, 20 steps): in_proj on the library vs as two half-width launches of the own kernel, and
# the text-side projections (y_embedder, batched K / V) on the library vs on the own kernel
for v in "0:0" "1:0" "1:1" "0:0" "1:0" "1:1"; do
  sp=${v%%:*}; tx=${v##*:}
  echo "== ZIGMA_IN_PROJ_SPLIT=$sp ZIGMA_TEXT_PROJ_OWN=$tx"
  ZIGMA_IN_PROJ_SPLIT=$sp ZIGMA_TEXT_PROJ_OWN=$tx python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
