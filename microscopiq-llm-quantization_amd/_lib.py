"""ctypes binding of libmsq_hip.so (include/msq.h).

The HIP library is the product path: if it is missing or a call fails this module
raises -- there is no CPU fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MSQ_LIB_OVERRIDE") or os.path.join(_HERE, "libmsq_hip.so")   # override: ablation builds (scripts/experiments)
_lib = None

MSQ_OK = 0
MSQ_ERR_BAD_ARG, MSQ_ERR_UNSUPPORTED, MSQ_ERR_LAUNCH = -1, -2, -3      # include/msq.h status codes
DTYPE_ID = {"torch.float32": 0, "torch.float16": 1, "torch.bfloat16": 2}

_i64, _i32, _f32, _vp = C.c_int64, C.c_int, C.c_float, C.c_void_p

_SIGS = {
    "msq_version": (C.c_int, []),
    "msq_last_error": (C.c_char_p, []),
    "msq_format_id": (C.c_int, [C.c_char_p]),
    "msq_format_params": (C.c_int, [_i32] + [C.POINTER(_i32)] * 3 + [C.POINTER(_f32)] * 2 + [C.POINTER(_i32)]),
    "msq_quantize_elemwise": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_quantize_format": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "msq_quantize_mx": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_quantize_mx_by_tile": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_quantize_mx_by_tile_py": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_quantize_mx_by_tile_ex": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp]),
    "msq_quantize_mx_lowp": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "msq_reduce_sum_inner": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "msq_reduce_max_inner": (C.c_int, [_vp, _vp, _i64, _i64, _vp]),
    "msq_outlier_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32]),
    "msq_outlier_fakequant": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i64,
                                        _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_vec_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_round": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_vec_gelu": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_add": (C.c_int, [_vp, _vp, _f32, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_rmsnorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_rmsnorm_mx_pack_a8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_vec_rmsnorm_mx_pack_a8_x16": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_vec_silu_mul_mx_pack_a8_x16": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_vec_silu": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_mul": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _vp]),
    "msq_vec_silu_mul_mx_pack_a8": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "msq_gptq_block_workspace_bytes": (_i64, [_i64, _i32]),
    "msq_gptq_block": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32,
                                 _f32, _i32, _i32, _vp]),
    "msq_kv_group_quant": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _i32, _i64, _i32, _vp]),
    "msq_floor_log2_lowp": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "msq_packed_kinds": (C.c_int, [_i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "msq_packed_sizes": (C.c_int, [_i64, _i64, _i32, _i32, _i32] + [C.POINTER(_i64)] * 4),
    "msq_packed_kinds_layout": (C.c_int, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "msq_outlier_pack": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                                   _f32, _i32, _i32, _i32, _i32, _vp]),
    "msq_pack_values": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "msq_outlier_unpack": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp]),
    "msq_qlinear_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "msq_qlinear_kernel_choice": (C.c_int, [_i64, _i64, _i64, _i32, _i32]),
    "msq_qlinear_kernel_name": (C.c_int, [_i64, _i64, _i64, _i32, _i32, _i32, C.c_char_p, _i32]),
    "msq_set_tuning": (C.c_int, [C.c_char_p, _i32]),
    "msq_qgemm256p_plan": (C.c_int, [_i64, _i64, _i64, _i32] + [C.POINTER(_i32)] * 4 + [C.POINTER(_i64)]),
    "msq_qgemm256p_segments": (C.c_int, [_i32] * 6 + [C.POINTER(_i32), _i32]),
    "msq_qlinear_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "msq_qlinear_f16x": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "msq_act_quant_workspace_bytes": (_i64, [_i64, _i64, _i32, _i32]),
    "msq_act_quant_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32,
                                     _i32, _i32, _vp]),
    "msq_act_quant_bf16_x16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _i32,
                                     _i32, _i32, _vp]),
    "msq_mx_pack_a8": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "msq_mx_pack_w4": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "msq_qlinear_mx_w4a8_workspace_bytes": (_i64, [_i64, _i64, _i64]),
    "msq_qlinear_mx_w4a8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp]),
    "msq_mx_pack_a8_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "msq_mx_pack_a8_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "msq_cast_f16_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "msq_read_stream_probe": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "msq_mx_pack_w8": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "msq_mx_pack_w6": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "msq_mx_pack_a6": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "msq_qlinear_mx_w6a8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _vp, _i64, _vp]),
    "msq_qlinear_mx_w8a8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp]),
    "msq_qlinear_w4a8_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32, _i32]),
    "msq_qlinear_w4a8": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "msq_qlinear_w4a8_x16": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
}


class MsqError(RuntimeError):
    """`rc` carries the C ABI's status code (include/msq.h: MSQ_ERR_*) when the error came from a library call, else None."""
    def __init__(self, *args, rc=None):
        super().__init__(*args)
        self.rc = rc


def so_path():
    return _SO


def lib():
    """Load libmsq_hip.so; raise loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise MsqError(
                f"{_SO} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C "
                f"{os.path.join(_HERE, 'csrc')}). There is no CPU fallback.")
        L = C.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)        # AttributeError if the symbol is missing: loud by design
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != MSQ_OK:
        msg = lib().msq_last_error()
        raise MsqError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}", rc=rc)


def format_id(name):
    fid = lib().msq_format_id(str(name).lower().encode())
    if fid < 0:
        raise Exception("Undefined elem format", name)      # formats.py:47
    return fid


def format_params(fmt_id):
    e, m, ex, kind = _i32(), _i32(), _i32(), _i32()
    mx, mn = _f32(), _f32()
    check(lib().msq_format_params(fmt_id, C.byref(e), C.byref(m), C.byref(ex), C.byref(mx), C.byref(mn),
                                  C.byref(kind)), "msq_format_params")
    return e.value, m.value, ex.value, mx.value, mn.value, kind.value


def ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    """raw hipStream_t of torch's current stream on `device` (the private accessor costs 0.3 us per call; building a
    torch.cuda.Stream object 4.5 us -- a third of the host time of a decode-size Linear)"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is None:
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    if isinstance(device, torch.device):
        idx = device.index
    elif isinstance(device, int):
        idx = device
    else:
        idx = None if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    return C.c_void_p(raw(idx))
