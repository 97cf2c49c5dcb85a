# same-box A/B of the training step (tools/train_probe.py, B = 64): forward projections and dX on the own kernels vs on the library
for rnd in 1 2; do
for v in 1 0; do
  echo -n "== ZIGMA_TRAIN_OWN_GEMMS=$v: "
  B=64 ZIGMA_TRAIN_OWN_GEMMS=$v python tools/train_probe.py 2>/dev/null | tail -1 | cut -c1-260
done
done
