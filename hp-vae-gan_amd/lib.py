"""ctypes binding of libhpvg.so (the C ABI declared in include/hpvg.h).

There is NO CPU fallback: every op of this package runs hand-written gfx950 kernels through this
library.  If the library is missing, or a tensor is not a contiguous fp32 device tensor, the call fails
loudly."""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("HPVG_LIB") or os.path.join(_HERE, "libhpvg.so")  # HPVG_LIB: development builds (tools/trace_conv.py)
HEADER_PATH = os.path.join(_ROOT, "include", "hpvg.h")

_lib = None

_ERR = {-1: "HPVG_ERR_ARG", -2: "HPVG_ERR_WORKSPACE", -3: "HPVG_ERR_UNSUPPORTED", -4: "HPVG_ERR_LAUNCH"}

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float
D = ctypes.c_double
Z = ctypes.c_size_t

# name -> argtypes (restype int unless listed in _SIZE_FUNCS)
_SIGS = {
    "hpvg_conv_wpack_floats": [I, I, I],
    "hpvg_conv_pack_weight_f32": [P, P, P, I, I, I, I, P],
    "hpvg_conv_pack_weight_batch_f32": [I, P, P, P, I, I, P],
    "hpvg_conv_wpack_floats_for": [I, I, I, I, I, I, I],
    "hpvg_conv_wants_wino2d": [I, I, I, I, I, I, I],
    "hpvg_conv_pack_weight_for_f32": [P, P, P, I, I, I, I, I, I, I, I, P],
    "hpvg_conv_pack_weight_batch_for_f32": [I, P, P, P, I, I, I, I, I, I, P],
    "hpvg_conv_fwd_ws_bytes": [I, I, I, I, I, I, I],
    "hpvg_conv_fwd_f32": [P, P, P, P, P, I, P, I, P, P, Z, I, I, I, I, I, I, I, P],
    "hpvg_conv_mask_words": [I, I, I, I, I],
    "hpvg_conv_fwd_bits_f32": [P, P, P, P, I, P, P, P, Z, I, I, I, I, I, I, I, P],
    "hpvg_conv_fwd_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_fwd_kernel_kind": [I, I, I, I, I, I, I],
    "hpvg_conv_bwd_weight_kernel_kind": [I, I, I, I, I, I, I],
    "hpvg_conv_narrow_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_wino_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_wino_config": [I, L],
    "hpvg_conv_bwd_weight_ws_bytes": [I, I, I, I, I, I, I],
    "hpvg_conv_bwd_weight_f32": [P, P, P, P, I, P, I, P, Z, I, I, I, I, I, I, I, P],
    "hpvg_conv_bwd_weight_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_bwd_weight_wino_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_bwd_weight_wino2_plan": [I, I, I, I, I, I, I, P],
    "hpvg_conv_bwd_weight_wino_config": [I],
    "hpvg_conv_bwd_weight_fuses_bias": [I, I, I, I, I, I, I],
    "hpvg_conv_bwd_weight_bias_f32": [P, P, P, I, P, I, P, Z, I, I, I, I, I, I, I, P],
    "hpvg_channel_sum_ws_bytes": [I],
    "hpvg_channel_sum_f32": [P, P, I, P, Z, I, I, L, P],
    "hpvg_bn_ws_bytes": [I],
    "hpvg_bn_train_stats_f32": [P, P, P, P, P, F, F, P, P, P, P, P, Z, I, I, L, P],
    "hpvg_affine_act_f32": [P, P, P, P, I, I, I, L, P],
    "hpvg_bn_train_fwd_f32": [P, P, P, P, P, F, F, P, P, P, P, P, I, I, P, Z, I, I, L, P],
    "hpvg_bn_act_bwd_f32": [P, P, P, P, P, P, I, I, P, P, P, I, P, Z, I, I, L, P],
    "hpvg_bn_bwd2_ws_bytes": [I],
    "hpvg_bn_act_bwd2_f32": [P, P, P, P, P, P, P, I, P, P, P, I, P, Z, I, I, L, P],
    "hpvg_bn_sums_f32": [P, P, P, Z, I, I, L, P],
    "hpvg_bn_finalize_f32": [P, D, P, P, P, P, F, F, P, P, P, P, I, P],
    "hpvg_bn_act_bwd_sums_f32": [P, P, P, P, P, P, I, P, P, Z, I, I, L, P],
    "hpvg_bn_act_bwd_apply_f32": [P, P, P, P, P, P, I, P, F, P, I, I, L, P],
    "hpvg_lrelu_mask_mul_f32": [P, P, P, L, P],
    "hpvg_add_f32": [P, P, P, L, P],
    "hpvg_copy_f32": [P, P, L, P],
    "hpvg_tanh_fwd_f32": [P, P, P, L, P],
    "hpvg_tanh_bwd_f32": [P, P, P, L, P],
    "hpvg_reparam_fwd_f32": [P, P, P, P, L, P],
    "hpvg_reparam_bwd_f32": [P, P, P, P, L, P],
    "hpvg_reduce_ws_bytes": [],
    "hpvg_kl_fwd_f32": [P, P, P, P, Z, L, P],
    "hpvg_kl_bwd_f32": [P, P, P, P, P, L, P],
    "hpvg_mse_fwd_f32": [P, P, P, P, Z, L, P],
    "hpvg_mse_bwd_f32": [P, P, P, P, L, P],
    "hpvg_sum_scaled_f32": [P, P, D, P, Z, L, P],
    "hpvg_sqsum_f32": [P, P, P, Z, L, P],
    "hpvg_fill_scaled_f32": [P, F, P, L, P],
    "hpvg_lerp_f32": [P, P, P, P, L, P],
    "hpvg_gp_fwd_f32": [P, P, F, P, Z, I, I, L, P],
    "hpvg_gp_bwd_f32": [P, P, P, F, I, I, L, P],
    "hpvg_upsample_linear_ac_f32": [P, P, P, F, P, L, I, I, I, I, I, I, P],
    "hpvg_normal_f32": [P, L, ctypes.c_ulonglong, ctypes.c_uint, P, P],
    "hpvg_uniform_f32": [P, L, ctypes.c_ulonglong, ctypes.c_uint, P, P],
    "hpvg_upsample_linear_ac_noise_f32": [P, P, P, F, L, I, I, I, I, I, I, I, I, ctypes.c_ulonglong, ctypes.c_uint, P, P],
    "hpvg_frames_resize_norm_u8_f32": [P, P, I, I, I, I, I, I, I, I, I, I, P],
    "hpvg_upsample_linear_ac_bwd_f32": [P, P, P, L, I, I, I, I, I, I, P],
    "hpvg_sn_power_iter_f32": [P, P, P, P, P, P, I, I, I, F, P, Z, P],
    "hpvg_div_scalar_f32": [P, P, P, L, P],
    "hpvg_sn_power_iter_batch_f32": [I, P, P, P, P, P, P, P, P, I, F, P, Z, P],
    "hpvg_sn_bwd_batch_f32": [I, P, P, P, P, P, P, P, P, P, Z, P],
    "hpvg_sn_bwd_ws_bytes": [I, I],
    "hpvg_sn_bwd_f32": [P, P, P, P, P, P, I, P, Z, I, I, P],
    "hpvg_clip_scale_f32": [P, L, P, F, P, P],
    "hpvg_adam_step_f32": [P, P, P, P, L, F, F, F, F, I, P, P],
    "hpvg_counter_inc_i32": [P, P],
    "hpvg_graph_node_census": [P, P, I],
    "hpvg_gate_fwd_f32": [P, P, P, P, I, I, L, P],
    "hpvg_gate_bwd_f32": [P, P, P, P, P, P, I, I, L, P],
    "hpvg_rowsum_f32": [P, P, P, F, I, I, L, P],
    "hpvg_outer_f32": [P, P, P, F, I, I, L, P],
    "hpvg_colsum_f32": [P, P, P, I, I, L, P],
    "hpvg_reparam_bern_fwd_f32": [P, P, P, L, P],
    "hpvg_reparam_bern_bwd_f32": [P, P, P, L, P],
    "hpvg_kl_bern_fwd_f32": [P, P, P, Z, L, P],
    "hpvg_kl_bern_bwd_f32": [P, P, P, L, P],
}
_SIZE_FUNCS = {"hpvg_conv_wpack_floats_for", "hpvg_conv_mask_words", "hpvg_bn_bwd2_ws_bytes", "hpvg_channel_sum_ws_bytes", "hpvg_conv_fwd_ws_bytes", "hpvg_conv_wpack_floats", "hpvg_conv_bwd_weight_ws_bytes", "hpvg_bn_ws_bytes", "hpvg_reduce_ws_bytes", "hpvg_sn_bwd_ws_bytes"}


def header_symbols():
    """Every function name declared in include/hpvg.h."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hpvg_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "hp-vae-gan_amd: %s not found. Build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "This package has no CPU fallback." % LIB_PATH)
    # torch has already loaded its HIP runtime (libamdhip64.so.7); libhpvg.so resolves against that same soname,
    # so stream handles and device pointers are shared with torch.
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = Z if name in _SIZE_FUNCS else I
    _lib = lib
    return lib


def check_symbols():
    lib = load()
    declared = header_symbols()
    missing = [s for s in declared if not hasattr(lib, s)]
    unbound = [s for s in declared if s not in _SIGS]
    if missing or unbound:
        raise ImportError("libhpvg.so / lib.py out of sync with include/hpvg.h: missing=%s unbound=%s" % (missing, unbound))
    return declared


def stream():
    """hipStream_t of torch's current stream (raw handle lookup: ~0.2 us, vs ~3 us for torch.cuda.current_stream())."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def ptr(t):
    """Device pointer of a contiguous fp32 device tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("hp-vae-gan_amd: tensor is on %s; these ops run only on an MI355X device (no CPU fallback)" % t.device)
    if t.dtype not in (torch.float32, torch.float64, torch.uint8, torch.int32):  # int32 also carries the 1-bit mask words
        raise RuntimeError("hp-vae-gan_amd: unsupported dtype %s" % t.dtype)
    if not t.is_contiguous():
        raise RuntimeError("hp-vae-gan_amd: tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _SIZE_FUNCS:
        return rc
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (name, _ERR.get(rc, rc)))
    return rc
