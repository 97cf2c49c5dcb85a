#!/bin/bash
# round 5, GPU call 1: new tests, the scan / in_proj / config-4 A/B against the round-4 library, forward A/B, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dt_in_kernel or dt_proj_in_kernel or other_state_sizes or linear_ws or small_batch_sequence" 2>&1 | tail -15 > gpurun_out/r05_c1_new_tests.txt
cat gpurun_out/r05_c1_new_tests.txt
python tools/r05_scan_ab.py 2>&1 | tail -3
fwd() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(label='$label', ms_per_step=d['ms_per_step'], scan_us=d['roofline']['launch_us'], check=d['check'])))" | tee -a gpurun_out/r05_c1_fwd_ab.jsonl
}
rm -f gpurun_out/r05_c1_fwd_ab.jsonl
for rep in 1 2; do
  fwd r4_lib ZIGMA_AMD_LIB=$R/tools/libzigma_base_r04.so
  fwd r5 X=1
  fwd r5_gate_in_in_proj ZIGMA_KNOBS=mamba_simple.GATE_IN_IN_PROJ=True
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_c1_gpu_tests_tail.txt
cat gpurun_out/r05_c1_gpu_tests_tail.txt
