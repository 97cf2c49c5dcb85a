# same-box A/B: MFMA results in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1) for conv_x_proj.hip / cross_attn_bwd.hip against hipcc's default (AGPRs)
# (tools/libzigma_agpr_cx.so / the default library: built in the container with zigma_amd.build.SOURCE_FLAGS set accordingly)
for rnd in 1 2 3; do
for v in vgpr agpr; do
  lib=""; [ $v = agpr ] && lib=$PWD/tools/libzigma_agpr_cx.so
  echo -n "== conv_x_proj / cross_attn_bwd MFMA results in $v: "
  ZIGMA_AMD_LIB=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['launch_us'], d['check']['rel_err_vs_unfused'])"
done
done
for v in vgpr agpr vgpr agpr; do
  lib=""; [ $v = agpr ] && lib=$PWD/tools/libzigma_agpr_cx.so
  echo -n "== train $v: "; B=64 ZIGMA_AMD_LIB=$lib python tools/train_probe.py 2>/dev/null | tail -1 | cut -c60-200
done
