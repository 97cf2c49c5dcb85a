"""ctypes binding of libzigma_hip.so (include/zigma_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel reports an error,
the calling op raises.  torch is used for device memory and streams only — every pointer crossing
the boundary is a raw device address (`Tensor.data_ptr()`), every launch goes to torch's current
HIP stream so that stream semantics (and graph capture) match the reference's extensions
(`at::cuda::getCurrentCUDAStream()`, selective_scan.cpp:326-327).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZIGMA_AMD_LIB") or os.path.join(_HERE, "lib", "libzigma_hip.so")   # env: A/B builds in tools/

F32, F16, BF16 = 0, 1, 2
_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}

i32, i64, vp, f32 = C.c_int32, C.c_int64, C.c_void_p, C.c_float


class ScanParams(C.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "dim", "seqlen", "dstate", "n_groups", "is_variable_B", "is_variable_C",
                            "delta_softplus", "io_dtype", "bc_dtype", "chunk_len", "flags")]
        + [(n, i64) for n in (
            "u_batch_stride", "u_d_stride", "u_l_stride",
            "delta_batch_stride", "delta_d_stride", "delta_l_stride",
            "z_batch_stride", "z_d_stride", "z_l_stride",
            "out_batch_stride", "out_d_stride", "out_l_stride",
            "out_z_batch_stride", "out_z_d_stride", "out_z_l_stride",
            "A_d_stride", "A_dstate_stride",
            "B_batch_stride", "B_group_stride", "B_d_stride", "B_dstate_stride", "B_l_stride",
            "C_batch_stride", "C_group_stride", "C_d_stride", "C_dstate_stride", "C_l_stride")]
        + [(n, vp) for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "z", "out", "out_z", "x",
                             "z_row_index", "out_row_index", "checkpoints")]
        + [("reset_period", i32), ("pad2_", i32), ("info", C.POINTER(C.c_int32))]
        + [("dt_x", vp), ("dt_w", vp), ("dt_x_batch_stride", i64), ("dt_x_l_stride", i64), ("dt_w_row_stride", i64),
           ("dt_rank", i32), ("pad3_", i32)]
    )


SCAN_Z_PREACTIVATED, SCAN_ACCUMULATE = 2, 4                # zigma_scan_params_t.flags
SCAN_PROBE_V1, SCAN_PROBE_PRIO_SHIFT, SCAN_PROBE_R5_SHIFT = 0x100, 9, 10           # A/B probes (tools/scan_ab.py, tools/r05_scan_ab.py)
SCAN_KERNEL_GENERIC, SCAN_KERNEL_TOK, SCAN_KERNEL_TOK2 = 1, 2, 3   # zigma_scan_params_t.info[0]


class ConvParams(C.Structure):
    _fields_ = (
        [(n, i32) for n in ("batch", "dim", "seqlen", "width", "silu_activation", "io_dtype", "w_dtype", "flags")]
        + [(n, i64) for n in ("x_batch_stride", "x_c_stride", "x_l_stride", "weight_c_stride", "weight_width_stride",
                              "out_batch_stride", "out_c_stride", "out_l_stride")]
        + [(n, vp) for n in ("x", "weight", "bias", "out", "x_row_index")]
        + [("reset_period", i32), ("pad2_", i32)]
    )


class NormParams(C.Structure):
    _fields_ = (
        [(n, i32) for n in ("rows", "cols", "rows_per_batch", "is_rms", "x_dtype", "res_dtype", "w_dtype", "mod_dtype")]
        + [("eps", f32), ("flags", i32)]
        + [(n, i64) for n in ("x_row_stride", "branch_row_stride", "x_out_row_stride", "res_row_stride",
                              "res_out_row_stride", "y_row_stride", "y_mod_row_stride", "mod_batch_stride")]
        + [(n, vp) for n in ("x", "branch", "gate", "x_out", "residual", "residual_out", "weight", "bias", "y_out",
                             "shift", "scale", "y_mod")]
    )


class DtProjParams(C.Structure):
    _fields_ = ([("m", i64), ("n", i32), ("k", i32), ("dtype", i32), ("softplus", i32), ("flags", i32), ("pad_", i32)]
                + [(n, i64) for n in ("x_row_stride", "w_row_stride", "out_row_stride")]
                + [(n, vp) for n in ("x", "w", "bias", "out")])


class ScanBwdParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "dim", "seqlen", "dstate", "delta_softplus", "io_dtype", "flags", "reset_period")]
                + [(n, i64) for n in (
                    "u_batch_stride", "u_l_stride", "delta_batch_stride", "delta_l_stride", "z_batch_stride", "z_l_stride",
                    "out_batch_stride", "out_l_stride", "dout_batch_stride", "dout_l_stride", "du_batch_stride",
                    "du_l_stride", "ddelta_batch_stride", "ddelta_l_stride", "dz_batch_stride", "dz_l_stride",
                    "A_d_stride", "A_dstate_stride", "B_batch_stride", "B_dstate_stride", "B_l_stride",
                    "C_batch_stride", "C_dstate_stride", "C_l_stride", "dB_batch_stride", "dB_dstate_stride",
                    "dB_l_stride", "dC_batch_stride", "dC_dstate_stride", "dC_l_stride")]
                + [(n, vp) for n in ("u", "delta", "A", "B", "C", "D", "delta_bias", "z", "out", "dout", "du", "ddelta",
                                     "dz", "dA", "dB", "dC", "dD", "ddelta_bias", "workspace")]
                + [("workspace_bytes", i64), ("z_row_index", vp), ("out_row_index", vp), ("checkpoints", vp)])


class ConvBwdParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "dim", "seqlen", "width", "silu_activation", "io_dtype", "w_dtype", "flags")]
                + [(n, i64) for n in ("x_batch_stride", "x_l_stride", "dout_batch_stride", "dout_l_stride", "dx_batch_stride",
                                      "dx_l_stride", "weight_c_stride", "weight_width_stride")]
                + [(n, vp) for n in ("x", "weight", "bias", "dout", "dx", "dweight", "dbias", "x_row_index", "workspace")]
                + [("workspace_bytes", i64), ("reset_period", i32), ("pad_", i32)])


class NormBwdParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("rows", "cols", "is_rms", "x_dtype", "res_dtype", "w_dtype")]
                + [("eps", f32), ("flags", i32)]
                + [(n, i64) for n in ("xsum_row_stride", "dy_row_stride", "dres_out_row_stride", "dx_row_stride",
                                      "dres_row_stride")]
                + [(n, vp) for n in ("xsum", "weight", "dy", "dresidual_out", "dx", "dresidual", "dweight", "dbias",
                                     "workspace")]
                + [("workspace_bytes", i64)])


class XAttnParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "seqlen", "n_ctx", "heads", "head_dim", "dtype", "flags")] + [("scale", f32)]
                + [(n, i64) for n in ("q_batch_stride", "q_row_stride", "k_batch_stride", "k_row_stride", "v_batch_stride",
                                      "v_row_stride", "o_batch_stride", "o_row_stride")]
                + [(n, vp) for n in ("q", "k", "v", "out")])


class XAttnBwdParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "seqlen", "n_ctx", "heads", "head_dim", "dtype", "flags")] + [("scale", f32)]
                + [("chunks", i32), ("pad_", i32)]
                + [(n, i64) for n in ("q_batch_stride", "q_row_stride", "k_batch_stride", "k_row_stride", "v_batch_stride",
                                      "v_row_stride", "do_batch_stride", "do_row_stride", "dq_batch_stride", "dq_row_stride")]
                + [(n, vp) for n in ("q", "k", "v", "dout", "dq", "dk_part", "dv_part")])


class PatchEmbedParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "in_chans", "height", "width", "patch", "embed_dim", "dtype", "flags")]
                + [(n, i64) for n in ("x_batch_stride", "x_chan_stride", "x_row_stride", "pos_row_stride", "out_batch_stride", "out_row_stride")]
                + [(n, vp) for n in ("x", "weight", "bias", "pos", "out")])


class TimestepEmbedParams(C.Structure):
    _fields_ = [("batch", i32), ("dim", i32), ("dtype", i32), ("flags", i32), ("out_row_stride", i64), ("t", vp), ("freqs", vp), ("out", vp)]


class FinalLayerParams(C.Structure):
    _fields_ = ([("rows", i64), ("cols", i32), ("n_out", i32), ("dtype", i32), ("flags", i32), ("eps", f32), ("pad_", i32),
                 ("x_row_stride", i64), ("out_row_stride", i64)] + [(n, vp) for n in ("x", "weight", "bias", "out")])


class SkinnyParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("m", "n", "k", "dtype", "flags", "pad_")] + [(n, i64) for n in ("x_row_stride", "w_row_stride", "out_row_stride")]
                + [(n, vp) for n in ("x", "w", "bias", "out")])


class GlueBwdParams(C.Structure):
    _fields_ = ([("rows", i64), ("cols", i32), ("rows_per_batch", i32), ("dtype", i32), ("flags", i32), ("s_add", f32), ("pad_", i32)]
                + [(n, i64) for n in ("dy_row_stride", "a_row_stride", "out_row_stride", "s_batch_stride")]
                + [(n, vp) for n in ("dy", "a", "s", "out", "r1", "r2")])


class CalibParams(C.Structure):
    _fields_ = [("mode", i32), ("iters", i32), ("bytes", i64), ("src", vp), ("dst", vp)]


class XProjParams(C.Structure):
    _fields_ = ([("m", i64), ("n", i32), ("k", i32), ("dtype", i32), ("flags", i32)]
                + [(n, i64) for n in ("x_row_stride", "w_row_stride", "out_row_stride")] + [(n, vp) for n in ("x", "w", "out")])


class ConvXProjParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "seqlen", "dim", "n", "dtype", "flags")]
                + [(n, i64) for n in ("x_batch_stride", "x_l_stride", "u_batch_stride", "u_l_stride", "w_row_stride", "out_row_stride")]
                + [(n, vp) for n in ("x", "conv_weight", "conv_bias", "w", "u", "out", "x_row_index")])


class LinearParams(C.Structure):
    _fields_ = ([("m", i64), ("n", i32), ("k", i32), ("dtype", i32), ("flags", i32), ("silu_from_col", i32), ("pad_", i32)]
                + [(n, i64) for n in ("x_row_stride", "w_row_stride", "out_row_stride")] + [(n, vp) for n in ("x", "w", "bias", "out")]
                + [("residual", vp), ("gate", vp), ("res_row_stride", i64), ("gate_batch_stride", i64), ("rows_per_batch", i32), ("pad2_", i32)])


EXPORTS = ("zigma_linear_fwd", "zigma_conv_x_proj_fwd", "zigma_scale_reduce_bwd", "zigma_selective_scan_fwd", "zigma_causal_conv1d_fwd", "zigma_add_norm_fwd", "zigma_dt_proj_softplus_fwd", "zigma_cross_attn_fwd", "zigma_x_proj_fwd", "zigma_selective_scan_bwd",
           "zigma_selective_scan_bwd_workspace_bytes", "zigma_causal_conv1d_bwd",
           "zigma_causal_conv1d_bwd_workspace_bytes", "zigma_add_norm_bwd", "zigma_add_norm_bwd_workspace_bytes",
           "zigma_strerror",
           "zigma_cross_attn_bwd", "zigma_cross_attn_bwd_chunks", "zigma_patch_embed_fwd", "zigma_timestep_embed_fwd", "zigma_final_layer_fwd",
           "zigma_skinny_linear_fwd", "zigma_calib_launch", "zigma_abi_version", "zigma_last_kernel")

_lib = None


def lib():
    """Load the shared library once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"zigma_amd: {LIB_PATH} not built — run `python -m zigma_amd.build` (hipcc, gfx950). "
                "There is no CPU or eager fallback for the HIP ops.")
        L = C.CDLL(LIB_PATH)
        for name, st in (("zigma_selective_scan_fwd", ScanParams), ("zigma_causal_conv1d_fwd", ConvParams),
                         ("zigma_add_norm_fwd", NormParams), ("zigma_dt_proj_softplus_fwd", DtProjParams),
                         ("zigma_selective_scan_bwd", ScanBwdParams), ("zigma_causal_conv1d_bwd", ConvBwdParams),
                         ("zigma_add_norm_bwd", NormBwdParams), ("zigma_cross_attn_fwd", XAttnParams), ("zigma_cross_attn_bwd", XAttnBwdParams), ("zigma_patch_embed_fwd", PatchEmbedParams),
                         ("zigma_timestep_embed_fwd", TimestepEmbedParams), ("zigma_final_layer_fwd", FinalLayerParams), ("zigma_skinny_linear_fwd", SkinnyParams), ("zigma_x_proj_fwd", XProjParams),
                         ("zigma_linear_fwd", LinearParams), ("zigma_conv_x_proj_fwd", ConvXProjParams), ("zigma_scale_reduce_bwd", GlueBwdParams), ("zigma_calib_launch", CalibParams)):
            fn = getattr(L, name)
            fn.argtypes = [C.POINTER(st), vp]
            fn.restype = C.c_int
        for name, st in (("zigma_selective_scan_bwd_workspace_bytes", ScanBwdParams),
                         ("zigma_causal_conv1d_bwd_workspace_bytes", ConvBwdParams),
                         ("zigma_add_norm_bwd_workspace_bytes", NormBwdParams)):
            fn = getattr(L, name)
            fn.argtypes = [C.POINTER(st)]
            fn.restype = C.c_int64
        L.zigma_cross_attn_bwd_chunks.argtypes = [C.c_int]
        L.zigma_cross_attn_bwd_chunks.restype = C.c_int
        L.zigma_strerror.argtypes = [C.c_int]
        L.zigma_strerror.restype = C.c_char_p
        L.zigma_abi_version.restype = C.c_int
        L.zigma_last_kernel.restype = C.c_char_p
        if L.zigma_abi_version() != 10:
            raise RuntimeError("zigma_amd: libzigma_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_kernel():
    return lib().zigma_last_kernel().decode()


def dtype_id(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"zigma_amd: unsupported dtype {t.dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("zigma_amd: HIP kernels need tensors on a GPU device (no CPU fallback); got a "
                               f"{t.device} tensor")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("zigma_amd: all tensors must be on the same device")
    return dev


def workspace(fn_name, params, device):
    """Allocate the caller-owned scratch buffer a backward entry point asks for and attach it to the parameter block."""
    nbytes = getattr(lib(), fn_name + "_workspace_bytes")(C.byref(params))
    ws = torch.empty(max(int(nbytes), 16), device=device, dtype=torch.uint8)
    params.workspace, params.workspace_bytes = ws.data_ptr(), nbytes
    return ws


TRACE = None        # tests / tools: set to a list to receive (entry point, kernel that served it, parameter block) per call


def call(fn_name, params, device):
    L = lib()
    with torch.cuda.device(device):
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = getattr(L, fn_name)(C.byref(params), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"{fn_name}: {L.zigma_strerror(rc).decode()} (status {rc})")
    if TRACE is not None:
        TRACE.append((fn_name, L.zigma_last_kernel().decode(), params))
