# same-box A/B: residual rows of the next block pair requested ahead of this pair's stores in linear4w's gated-residual epilogue (the default)
# against one pair at a time (tools/libzigma_res_noahead.so: built in the container from the generator with L4W_RES_AHEAD=0)
for v in ahead noahead ahead noahead; do
  lib=""; [ $v = noahead ] && lib=$PWD/tools/libzigma_res_noahead.so
  echo "== $v (stand-alone epilogue probe):"
  ZIGMA_AMD_LIB=$lib python tools/linear4w_epi_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  ', d.get('shape'), {k: round(v, 1) for k, v in d.get('us', {}).items()})"
done
for rnd in 1 2 3; do
for v in ahead noahead; do
  lib=""; [ $v = noahead ] && lib=$PWD/tools/libzigma_res_noahead.so
  echo -n "== $v: "
  ZIGMA_AMD_LIB=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
done
