for rnd in 1 2 3 4; do
for v in base l4w; do
  lib=""; [ $v != base ] && lib=$PWD/tools/libzigma_nt_$v.so   # built in the container: the generator with L4W_STORE_NT=0 or 1, then zigma_amd.build(lib=...)
  echo -n "== $v: "
  ZIGMA_AMD_LIB=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'])"
done
done
