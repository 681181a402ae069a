"""ctypes binding of ``libsiglip_hip.so`` (C ABI declared in ``include/siglip_hip.h``).

There is no CPU fallback: if the library is missing, or lacks any symbol the header declares, ``load()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGL_LIB_PATH") or os.path.join(_HERE, "libsiglip_hip.so")  # override: developer A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "siglip_hip.h")

SGL_DTYPE_F32, SGL_DTYPE_BF16, SGL_DTYPE_BF16X3 = 0, 1, 2
EPI_STORE, EPI_BIAS_GELU, EPI_RES_F32, EPI_QKV, EPI_GELU_BWD, EPI_POS_F32, EPI_F32 = range(7)
STATUS = {0: "ok", -1: "bad shape", -2: "unsupported configuration", -3: "buffer too small", -4: "HIP error",
          -5: "null pointer"}

_fp = C.c_void_p  # device pointers travel as integers


class SglConfig(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_layers", C.c_int),
                ("num_heads", C.c_int), ("patch_size", C.c_int), ("native_grid", C.c_int),
                ("layer_norm_eps", C.c_float), ("compute_dtype", C.c_int), ("use_head", C.c_int)]


LAYER_FIELDS = ["ln1_w", "ln1_b", "q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b", "ln2_w", "ln2_b",
                "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
HEAD_FIELDS = ["probe", "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "head_ln_w", "head_ln_b",
               "head_fc1_w", "head_fc1_b", "head_fc2_w", "head_fc2_b"]


class SglLayerPtrs(C.Structure):
    _fields_ = [(n, _fp) for n in LAYER_FIELDS]


class SglWeights(C.Structure):
    _fields_ = ([("patch_w", _fp), ("patch_b", _fp), ("pos", _fp), ("layers", C.POINTER(SglLayerPtrs)),
                 ("post_ln_w", _fp), ("post_ln_b", _fp)] + [(n, _fp) for n in HEAD_FIELDS])


class SglGrads(C.Structure):
    _fields_ = ([("patch_w", _fp), ("patch_b", _fp), ("pos", _fp), ("layers", C.POINTER(SglLayerPtrs)),
                 ("post_ln_w", _fp), ("post_ln_b", _fp)] + [(n, _fp) for n in HEAD_FIELDS] +
                [("accumulate", C.c_int)])


class SglAdamwTensor(C.Structure):
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("n", C.c_uint64), ("lr", C.c_float),
                ("weight_decay", C.c_float)]


class SglAdamwAux(C.Structure):
    _fields_ = [("dst", _fp), ("dst_t", _fp), ("dst_f32", _fp), ("ema", _fp), ("ld", C.c_int), ("ld_t", C.c_int),
                ("rows", C.c_int), ("cols", C.c_int), ("row0", C.c_int), ("dtype", C.c_int), ("group", C.c_int),
                ("reserved", C.c_int)]


_lib = None


def declared_symbols() -> list[str]:
    """Every function name the public header declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgl_[a-z0-9_]+)\s*\(", text)))


def _sig(lib, name, restype, argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


def load():
    """Load the HIP library (once).  Raises RuntimeError when it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the encoder path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks symbols declared in siglip_hip.h: {missing}")
    i, sz, f = C.c_int, C.c_size_t, C.c_float
    psz = C.POINTER(C.c_size_t)
    _sig(lib, "sgl_abi_version", i, [])
    _sig(lib, "sgl_status_string", C.c_char_p, [i])
    _sig(lib, "sgl_create", C.c_void_p, [C.POINTER(SglConfig)])
    _sig(lib, "sgl_destroy", None, [C.c_void_p])
    _sig(lib, "sgl_last_hip_error", i, [C.c_void_p])
    _sig(lib, "sgl_query_sizes", i, [C.c_void_p, i, i, i, i, psz, psz, psz])
    _sig(lib, "sgl_prepare_weights", i, [C.c_void_p, C.POINTER(SglWeights), _fp, sz, _fp])
    _sig(lib, "sgl_prepare_weights_dirty", i, [C.c_void_p, C.POINTER(SglWeights), _fp, sz, C.c_char_p, i, _fp])
    _sig(lib, "sgl_forward", i, [C.c_void_p, C.POINTER(SglWeights), _fp, _fp, i, i, i, i, i, _fp, i, _fp, _fp, _fp, sz,
                                 _fp, sz, _fp])
    _sig(lib, "sgl_forward_ex", i, [C.c_void_p, C.POINTER(SglWeights), _fp, _fp, i, i, i, i, i, _fp, i, _fp, _fp, _fp, sz,
                                    _fp, sz, i, _fp])
    _sig(lib, "sgl_forward_slots", i, [C.c_void_p, C.POINTER(SglWeights), _fp, _fp, i, i, i, i, i, C.POINTER(_fp), _fp, _fp,
                                       _fp, sz, _fp, sz, i, _fp])
    _sig(lib, "sgl_backward_begin_p", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglGrads), i, i, i, _fp, _fp,
                                          _fp, _fp, _fp, sz, _fp, sz, _fp])
    _sig(lib, "sgl_backward_layer_p", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglGrads), i, i, i, i, _fp,
                                          _fp, i, _fp, sz, _fp, sz, _fp])
    _sig(lib, "sgl_backward_begin", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglGrads), i, i, i, _fp, _fp,
                                        _fp, _fp, _fp, sz, _fp, sz, _fp])
    _sig(lib, "sgl_backward_layer", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglGrads), i, i, i, i, _fp,
                                        _fp, i, _fp, sz, _fp, sz, _fp])
    _sig(lib, "sgl_backward_embed", i, [C.c_void_p, C.POINTER(SglWeights), C.POINTER(SglGrads), i, i, i, i, _fp, sz,
                                        _fp, sz, _fp])
    _sig(lib, "sgl_backward", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglGrads), i, i, i, i, _fp,
                                  C.POINTER(_fp), _fp, _fp, i, i, _fp, sz, _fp, sz, _fp])
    _sig(lib, "sgl_op_layernorm_fwd", i, [_fp, _fp, _fp, _fp, i, _fp, _fp, i, i, f, _fp])
    _sig(lib, "sgl_op_layernorm_bwd", i, [_fp, i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, i, _fp, _fp, _fp, sz, i, i, _fp])
    _sig(lib, "sgl_op_gemm_nt", i, [i, _fp, i, _fp, i, i, i, i, i, _fp, i, _fp, i, _fp, _fp, i, _fp, i, _fp, i, i, i,
                                    i, i, i, _fp])
    _sig(lib, "sgl_op_gemm_tn", i, [i, _fp, i, _fp, i, i, i, i, i, _fp, i, i, _fp])
    _sig(lib, "sgl_op_gemm_tn_ws", i, [i, _fp, i, _fp, i, i, i, i, i, _fp, i, i, _fp, sz, _fp])
    _sig(lib, "sgl_op_attn_fwd", i, [i, _fp, _fp, _fp, _fp, _fp, i, i, i, i, i, i, _fp])
    _sig(lib, "sgl_op_attn_bwd", i, [i, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, i, i, i, i, i, i, _fp])
    _sig(lib, "sgl_op_colsum", i, [i, _fp, i, i, i, _fp, i, _fp, sz, _fp])
    _sig(lib, "sgl_op_im2col", i, [_fp, i, _fp, i, i, i, i, i, i, _fp])
    _sig(lib, "sgl_op_pos_resize", i, [_fp, i, _fp, i, i, i, _fp])
    i64 = C.c_int64
    _sig(lib, "sgl_adamw_plan", i64, [C.POINTER(C.c_uint64), i, C.POINTER(C.c_int32), i64])
    _sig(lib, "sgl_op_grad_norm", i, [_fp, _fp, i64, f, _fp, _fp, _fp])
    _sig(lib, "sgl_op_adamw", i, [_fp, _fp, i64, C.c_double, C.c_double, C.c_double, i, _fp, _fp])
    _sig(lib, "sgl_op_ema", i, [_fp, _fp, i64, C.c_double, _fp])
    _sig(lib, "sgl_adamw_bind_shadows", i, [C.c_void_p, C.POINTER(SglWeights), _fp, C.POINTER(SglAdamwTensor),
                                            C.POINTER(SglAdamwAux), i])
    _sig(lib, "sgl_op_grad_norm_scaled", i, [_fp, _fp, i64, f, f, _fp, _fp, _fp])
    _sig(lib, "sgl_op_adamw_ex", i, [_fp, _fp, _fp, i64, C.c_double, C.c_double, C.c_double, i, _fp,
                                     C.POINTER(C.c_float), i, C.c_double, _fp])
    _sig(lib, "sgl_op_dwconv3x3", i, [_fp, i, _fp, _fp, _fp, i, i, i, i, i, _fp])
    _sig(lib, "sgl_op_dwconv3x3_wgrad_scratch_bytes", sz, [i, i, i, i])
    _sig(lib, "sgl_op_dwconv3x3_wgrad", i, [_fp, _fp, i, _fp, i, _fp, sz, i, i, i, i, _fp])
    _sig(lib, "sgl_op_preprocess", i, [_fp, i, i, i, i, _fp, i, i, i, i, i, f, f, _fp, f, _fp])
    _sig(lib, "sgl_op_preprocess_aug", i, [_fp, i, i, i, i, _fp, i, i, i, i, i, f, f, _fp, _fp, _fp])
    _sig(lib, "sgl_op_l2norm_tmean_fwd", i, [_fp, _fp, _fp, i, i, i, _fp])
    _sig(lib, "sgl_op_l2norm_tmean_bwd", i, [_fp, _fp, _fp, _fp, i, i, i, _fp])
    _sig(lib, "sgl_op_gate_mul", i, [_fp, _fp, _fp, sz, i, _fp])
    _sig(lib, "sgl_op_gate_mul_bwd", i, [_fp, _fp, _fp, _fp, _fp, sz, i, _fp])
    _sig(lib, "sgl_op_seg_loss_chunks", i, [i])
    _sig(lib, "sgl_op_seg_loss_fwd", i, [_fp, _fp, _fp, i, i, i, _fp])
    _sig(lib, "sgl_op_seg_loss_bwd", i, [_fp, _fp, _fp, _fp, _fp, i, i, i, f, _fp])
    _lib = lib
    return lib


class SglError(RuntimeError):
    pass


def check(status: int, what: str, ctx=None):
    if status == 0:
        return
    msg = STATUS.get(status, f"status {status}")
    if status == -4 and ctx is not None and _lib is not None:
        msg += f" (hipError_t {_lib.sgl_last_hip_error(ctx)})"
    raise SglError(f"{what}: {msg}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream_handle() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
