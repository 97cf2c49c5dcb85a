"""The step right after the ODE ON THE DEVICE (reference sample_acc.py:319-321,363-392,435): decode ->
uint8 -> gather, with a stand-in decoder of the VAE's shape behaviour (4 -> 3 channels, 8x upsampling)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cpu_formulas(latents, dec_cpu, is_video):
    """sample_acc.py:363-377 + :319-321 evaluated on the CPU in float64"""
    lat = latents.double().cpu()
    if not is_video:
        img = dec_cpu(lat / 0.18215)
    else:
        img = torch.stack([dec_cpu(lat[i] / 0.18215) for i in range(len(lat))], dim=1)
    return img, torch.clamp(127.5 * img + 128.0, 0, 255).to(torch.uint8)


@pytest.mark.parametrize("is_video", [False, True])
def test_finish_samples_on_device(is_video):
    from zigma_amd import postprocess as pp
    torch.manual_seed(3)
    lat = torch.randn((3, 5, 4, 16, 16) if is_video else (6, 4, 32, 32), device=DEV) * 0.18215 * 1.5
    dec = pp.StandInDecoder(seed=1, device=DEV)
    dec_cpu = pp.StandInDecoder(seed=1, dtype=torch.float64)
    got = pp.finish_samples(lat, dec, is_video=is_video, world=1)
    assert got.is_cuda and got.dtype == torch.uint8
    assert got.shape == ((5, 3, 3, 128, 128) if is_video else (6, 3, 256, 256))       # video: (T, B, 3, H, W), sample_acc.py:370-377
    # (1) the uint8 step is bit-exact: the device's conversion of the device's own decoded images == the CPU's conversion of them
    img_dev = pp.decode_latents(lat, dec, is_video=is_video)
    assert torch.equal(got.cpu(), pp.to_uint8(img_dev.cpu()))
    assert torch.equal(got.cpu(), torch.clamp(127.5 * img_dev.cpu() + 128.0, 0, 255).to(torch.uint8))
    # (2) the whole step against the CPU formulas in float64: images to fp32 rounding, pixels within one level (a level flips only
    # where 127.5 x + 128 lies within fp32 rounding of an integer)
    img_ref, u8_ref = _cpu_formulas(lat, dec_cpu, is_video)
    assert float((img_dev.double().cpu() - img_ref).abs().max()) < 2e-5
    d = (got.cpu().to(torch.int16) - u8_ref.to(torch.int16)).abs()
    assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3
    # pixel-space models: no decode
    x = torch.rand(2, 3, 8, 8, device=DEV) * 2 - 1
    assert torch.equal(pp.finish_samples(x, None, world=1), pp.to_uint8(x))
    # saturation at both ends
    assert pp.to_uint8(torch.tensor([-3.0, -1.0, 0.0, 0.996, 1.0, 7.0], device=DEV)).tolist() == [0, 0, 128, 254, 255, 255]


def test_ode_then_finish_samples_end_to_end():
    """sample_ode (euler, 4 steps) on a small HIP model, then decode -> uint8 on the device: the path of sample_acc.py:362-392"""
    from zigma_amd import postprocess as pp
    from zigma_amd.model_zigma import ZigMa
    from zigma_amd.transport import Sampler, create_transport
    torch.manual_seed(0)
    m = ZigMa(in_channels=4, embed_dim=64, depth=4, img_dim=8, patch_size=1, scan_type="zigzagN8", use_pe=2, device=DEV).eval()
    with torch.no_grad():
        for blk in m.blocks:
            blk.adaLN_modulation[-1].bias.normal_(std=0.3)
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=4)
    z = torch.randn(3, 4, 8, 8, device=DEV)
    with torch.no_grad():
        lat = fn(z, m.forward)[-1]
    img = pp.finish_samples(lat, pp.StandInDecoder(device=DEV), world=1)
    assert img.shape == (3, 3, 64, 64) and img.dtype == torch.uint8 and img.is_cuda
    assert 0 < int(img.min()) or int(img.max()) < 255 or img.float().std() > 0      # not constant
