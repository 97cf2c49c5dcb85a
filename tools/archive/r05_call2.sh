#!/bin/bash
# round 5, GPU call 2: d16_hi loads in the scan — targeted tests, scan A/B vs the round-4 library, forward A/B, the shapes probe, the whole suite
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan or small_batch or mamba_inner" 2>&1 | tail -12 > gpurun_out/r05_c2_scan_tests.txt
cat gpurun_out/r05_c2_scan_tests.txt
python tools/r05_scan_ab.py 2>&1 | tail -2
fwd() {
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(label='$label', ms_per_step=d['ms_per_step'], scan_us=d['roofline']['launch_us'], check=d['check']['rel_err_vs_unfused'])))" | tee -a gpurun_out/r05_c2_fwd_ab.jsonl
}
rm -f gpurun_out/r05_c2_fwd_ab.jsonl
for rep in 1 2 3; do
  fwd r4_lib ZIGMA_AMD_LIB=$R/tools/libzigma_base_r04.so
  fwd r5 X=1
done
python tools/r05_shapes_probe.py 2>&1 | tail -30
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_c2_gpu_tests_tail.txt
cat gpurun_out/r05_c2_gpu_tests_tail.txt
