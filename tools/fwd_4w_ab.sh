for v in "auto:1024" "auto:0" "auto:4096" ; do
  pol=${v%%:*}; mx=${v##*:}
  echo "== ZIGMA_LINEAR=$pol ZIGMA_4W_MAX_N=$mx"
  ZIGMA_LINEAR=$pol ZIGMA_4W_MAX_N=$mx python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['check'])"
done
