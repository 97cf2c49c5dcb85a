#!/bin/bash
# round 5, GPU call 4: weight-stationary 128-feature panels (out_proj below the 4-wave floor), probes, B=8/16 policy A/B, whole suite
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "x_proj or bench_block_path or linear_ws or no_text" 2>&1 | tail -6 > gpurun_out/r05_c4_tests.txt
cat gpurun_out/r05_c4_tests.txt
python tools/r05_shapes_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-900
python tools/r05_scan_ab.py 2>&1 | tail -1
fwd() {
  local label=$1 b=$2; shift; shift
  env "$@" python bench.py --no-cpu-baseline --batch $b --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(dict(label='$label', batch=$b, ms_per_step=d['ms_per_step'], scan_us=(d['roofline'] or {}).get('launch_us'), check=d['check']['rel_err_vs_unfused'])))" | tee -a gpurun_out/r05_c4_small_batch_ab.jsonl
}
rm -f gpurun_out/r05_c4_small_batch_ab.jsonl
for rep in 1 2; do for b in 8 16; do
  fwd r4_lib_r4_policy $b ZIGMA_AMD_LIB=$R/tools/libzigma_base_r04.so ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FUSE_NEEDS_4W=False,model_zigma.TO_Q_WS_MAX_TOKENS=0,model_zigma.TO_Q_OWN_MIN_TOKENS=1000000000,mamba_simple.OUT_PROJ_WS_MAX_TOKENS=0
  fwd r5_lib_r4_policy $b ZIGMA_KNOBS=mamba_simple.OUT_PROJ_FUSE_NEEDS_4W=False,model_zigma.TO_Q_WS_MAX_TOKENS=0,model_zigma.TO_Q_OWN_MIN_TOKENS=1000000000,mamba_simple.OUT_PROJ_WS_MAX_TOKENS=0
  fwd r5_out_proj_library $b ZIGMA_KNOBS=mamba_simple.OUT_PROJ_WS_MAX_TOKENS=0
  fwd r5_no_toq_ws $b ZIGMA_KNOBS=model_zigma.TO_Q_WS_MAX_TOKENS=0
  fwd r5 $b X=1
done; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05_c4_gpu_tests_tail.txt
cat gpurun_out/r05_c4_gpu_tests_tail.txt
