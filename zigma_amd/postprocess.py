"""The step right after the ODE: latent -> image decode, uint8 conversion, gather.

Mirrors `sample_acc.py:362-392,435-436` of the reference:

    samples = sample_fn(z, model_fn, **kw)[-1]
    samples = vae.decode(samples / 0.18215).sample                 (is_latent; per sample + stack(dim=1) for video)
    sam_4fid = clamp(127.5 * samples + 128.0, 0, 255).to(uint8)    (samples_pil, :319)
    sam_4fid = accelerator.gather(sam_4fid)                        (:435)

The VAE is third-party code (`diffusers.AutoencoderKL`, absent here) and stays pluggable: `decode` is any callable
latents -> images in [-1, 1] (e.g. `lambda z: vae.decode(z).sample`); `None` means the model works in pixel space.
Everything stays on the device; the gather is one `all_gather_into_tensor` of uint8 pixels (RCCL on GPUs)."""
import torch

from . import sharded_sampling as ss

LATENT_SCALE = 0.18215          # Stable-Diffusion VAE scaling used by the reference (sample_acc.py:365)


def to_uint8(images):
    """samples_pil (sample_acc.py:319-321): clamp(127.5 x + 128, 0, 255) truncated to uint8."""
    return torch.clamp(127.5 * images + 128.0, 0, 255).to(torch.uint8)


def decode_latents(latents, decode=None, latent_scale=LATENT_SCALE, is_video=False):
    """Images in [-1, 1] from the ODE's final state.  Video latents (B, T, C, H, W) are decoded sample by sample and
    stacked along dim 1 exactly like the reference does (sample_acc.py:370-377: the result is (T, B, 3, H', W'))."""
    if decode is None:
        return latents
    if not is_video:
        return decode(latents / latent_scale)
    return torch.stack([decode(latents[i] / latent_scale) for i in range(len(latents))], dim=1)


class StandInDecoder(torch.nn.Module):
    """A fixed, seeded latent -> image map with the SHAPE behaviour of the reference's VAE decoder (4 latent channels -> 3 image
    channels, 8x upsampling: `vae.decode(z).sample`, sample_acc.py:365) for tests and end-to-end timing where the third-party
    `diffusers.AutoencoderKL` is absent: an 8x8 transposed conv (stride 8) + tanh.  NOT a VAE — results are not images."""

    def __init__(self, latent_channels=4, image_channels=3, up=8, seed=0, device=None, dtype=torch.float32):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        w = torch.randn(latent_channels, image_channels, up, up, generator=g) * (0.5 / latent_channels ** 0.5)
        self.up = up
        self.register_buffer("weight", w.to(device=device, dtype=dtype))
        self.register_buffer("bias", (torch.randn(image_channels, generator=g) * 0.1).to(device=device, dtype=dtype))

    def forward(self, z):
        return torch.tanh(torch.nn.functional.conv_transpose2d(z.to(self.weight.dtype), self.weight, self.bias, stride=self.up))


def finish_samples(latents, decode=None, latent_scale=LATENT_SCALE, is_video=False, world=None):
    """decode -> uint8 -> gather over ranks (dim 0, rank order).  Returns a uint8 tensor on the latents' device."""
    with torch.no_grad():
        img = to_uint8(decode_latents(latents, decode, latent_scale, is_video))
    if world is None:
        world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    return ss.gather_samples(img, world)
