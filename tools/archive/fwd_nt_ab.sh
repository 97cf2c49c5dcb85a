# same-box A/B inside the forward (bench.py, 20 steps): the weight-stationary in_proj kernel with streaming (nt) output stores (the default)
# against the default cache policy (tools/libzigma_nont.so: built in the container with -DZIGMA_WS_NO_NT), and against in_proj as two
# half-width launches of the tiled kernel
for rnd in 1 2 3; do
for v in "nt:1" "nont:1" "nt:0"; do
  l=${v%%:*}; ws=${v##*:}
  lib=""; [ $l = nont ] && lib=$PWD/tools/libzigma_nont.so
  echo -n "== linear_ws stores $l, ZIGMA_IN_PROJ_WS=$ws: "
  ZIGMA_AMD_LIB=$lib ZIGMA_IN_PROJ_WS=$ws python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
done
