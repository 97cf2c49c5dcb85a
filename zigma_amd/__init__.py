"""zigma_amd — MI355X-native implementation of ZigMa's denoiser-forward / ODE-sampling hot path.

Layout (mirrors the reference's module names so call sites read the same):
    csrc/ + include/zigma_hip.h      hand-written HIP kernels (gfx950) behind a C ABI
    _lib                             ctypes binding (raw pointers + stream)
    selective_scan_interface         selective_scan_fn, mamba_inner_fn, ...      (dis_mamba/.../selective_scan_interface.py)
    causal_conv1d_interface          causal_conv1d_fn                            (dis_causal_conv1d/.../causal_conv1d_interface.py)
    layernorm                        rms_norm_fn, layer_norm_fn, RMSNorm         (dis_mamba/.../triton/layernorm.py)
    mamba_simple                     Mamba                                       (dis_mamba/.../modules/mamba_simple.py)
    model_zigma                      ZigMa, Block, ...                           (model_zigma.py)
    scan_paths                       zigzag_path, hilbert_path, reverse_permut_np (utils/utils_zigzag.py)
    transport                        create_transport, Sampler                   (transport/)
    extension_shims                  `selective_scan_cuda` / `causal_conv1d_cuda` module stand-ins
"""
__version__ = "0.1.0"
