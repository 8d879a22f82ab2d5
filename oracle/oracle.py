"""ctypes front-end of the CPU oracle (oracle/msq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
Parity status: pinned by tests/test_oracle_golden.py against tests/golden/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RD = {"nearest": 0, "floor": 1, "even": 2}
VARIANT = {"quant": 0, "mx_ops": 1}


def build(force=False):
    so = os.path.join(_HERE, "libmsq_oracle.so")
    src = os.path.join(_HERE, "msq_oracle.c")
    if force or (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.msq_oracle_posit_decode.restype = C.c_double
        _LIB.msq_oracle_posit_decode.argtypes = [C.c_uint32, C.c_int, C.c_int]
        _LIB.msq_oracle_posit_encode.restype = C.c_uint32
        _LIB.msq_oracle_posit_encode.argtypes = [C.c_double, C.c_int, C.c_int]
    return _LIB


def set_threads(n=0):
    """OpenMP threads of the fake-quant / linear loops (0 = leave as is); returns the count in effect."""
    return int(lib().msq_oracle_set_threads(int(n)))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def format_params(name):
    e, m, ex, kind = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    mx, mn = C.c_float(), C.c_float()
    rc = lib().msq_oracle_format_params(name.lower().encode(), C.byref(e), C.byref(m), C.byref(ex),
                                        C.byref(mx), C.byref(mn), C.byref(kind))
    if rc != 0:
        raise Exception("Undefined elem format", name)
    return e.value, m.value, ex.value, mx.value, mn.value, kind.value


def _pap(shape, axis):
    shape = tuple(int(s) for s in shape)
    axis = axis % len(shape)
    pre = int(np.prod(shape[:axis], dtype=np.int64))
    post = int(np.prod(shape[axis + 1:], dtype=np.int64))
    return pre, shape[axis], post


def quantize_elemwise_core(a, bits, exp_bits, max_norm, round="nearest", saturate_normals=False,
                           allow_denorm=True, bitwise=False):
    a = _f32(a)
    out = np.empty_like(a)
    fn = lib().msq_oracle_quantize_elemwise_bits if bitwise else lib().msq_oracle_quantize_elemwise_core
    fn(_p(a), _p(out), C.c_int64(a.size), C.c_int(bits), C.c_int(exp_bits), C.c_float(max_norm),
       C.c_int(RD[round]), C.c_int(bool(saturate_normals)), C.c_int(bool(allow_denorm)))
    return out


def outlier_fakequant(a, inlier_scale_bits, outlier_scale_bits, inlier_fmt, outlier_fmt, std_dev=2,
                      axis=0, block_size=0, round="nearest", flush_fp32_subnorms=False,
                      variant="quant", want_num_outliers=False):
    """Returns dict(out, mask, e_in, e_out, status[, num_outliers])."""
    a = _f32(a)
    pre, axis_len, post = _pap(a.shape, axis)
    blk = block_size if block_size > 0 else axis_len
    nblk = (axis_len + blk - 1) // blk
    out = np.empty_like(a)
    mask = np.empty(a.shape, dtype=np.uint8)
    e_in = np.empty((pre, nblk, post), dtype=np.float32)
    e_out = np.empty((pre, nblk, post), dtype=np.float32)
    n_out = None
    if want_num_outliers:
        assert pre == 1
        n_out = np.zeros((((nblk + blk - 1) // blk) * post,), dtype=np.int8)
    st = lib().msq_oracle_outlier_fakequant(
        _p(a), _p(out), _p(mask, C.c_uint8), _p(e_in), _p(e_out), _p(n_out, C.c_int8),
        C.c_int64(pre), C.c_int64(axis_len), C.c_int64(post), C.c_int(blk),
        inlier_fmt.lower().encode(), outlier_fmt.lower().encode(), C.c_int(inlier_scale_bits),
        C.c_int(outlier_scale_bits), C.c_double(float(std_dev)), C.c_int(RD[round]),
        C.c_int(bool(flush_fp32_subnorms)), C.c_int(VARIANT[variant]))
    if st < 0:
        raise Exception("Undefined elem format", inlier_fmt, outlier_fmt)
    r = dict(out=out, mask=mask, e_in=e_in, e_out=e_out, status=st)
    if want_num_outliers:
        r["num_outliers"] = n_out
    return r


LOWP = {"float16": 1, "fp16": 1, "f16": 1, "bfloat16": 2, "bf16": 2}


def outlier_fakequant_lowp(a, dtype, inlier_scale_bits, outlier_scale_bits, inlier_fmt, outlier_fmt, std_dev=2,
                           axis=0, block_size=0, round="nearest", flush_fp32_subnorms=False):
    """utils/quant.py:147-266 computed IN the tensor dtype (fp16 / bf16, every op rounded back as ATen's CPU half
    kernels do): `a` holds the tensor's values as float32.  Returns dict(out, mask, e_in, e_out, status)."""
    a = _f32(a)
    pre, axis_len, post = _pap(a.shape, axis)
    blk = block_size if block_size > 0 else axis_len
    nblk = (axis_len + blk - 1) // blk
    out = np.empty_like(a)
    mask = np.empty(a.shape, dtype=np.uint8)
    e_in = np.empty((pre, nblk, post), dtype=np.float32)
    e_out = np.empty((pre, nblk, post), dtype=np.float32)
    st = lib().msq_oracle_outlier_fakequant_lowp(
        _p(a), _p(out), _p(mask, C.c_uint8), _p(e_in), _p(e_out), C.c_int(LOWP[str(dtype).replace("torch.", "")]),
        C.c_int64(pre), C.c_int64(axis_len), C.c_int64(post), C.c_int(blk), inlier_fmt.lower().encode(),
        outlier_fmt.lower().encode(), C.c_int(inlier_scale_bits), C.c_int(outlier_scale_bits),
        C.c_double(float(std_dev)), C.c_int(RD[round]), C.c_int(bool(flush_fp32_subnorms)))
    if st == -1:
        raise Exception("Undefined elem format", inlier_fmt, outlier_fmt)
    if st == -2:
        raise Exception("lowp oracle: fp16 / bf16 with non-posit formats only")
    return dict(out=out, mask=mask, e_in=e_in, e_out=e_out, status=st)


def floor_log2_lowp(v, dtype):
    v = _f32(v)
    out = np.empty_like(v)
    lib().msq_oracle_floor_log2_lowp(_p(v), _p(out), C.c_int64(v.size), C.c_int(LOWP[str(dtype).replace("torch.", "")]))
    return out


def floor_log2_f32(v):
    """floor(torch.log2(v)) of a float32 tensor as the reference's Python path computes it (utils/quant.py:525-529)."""
    v = _f32(v)
    out = np.empty_like(v)
    lib().msq_oracle_floor_log2_f32(_p(v), _p(out), C.c_int64(v.size))
    return out


def round_lowp(v, dtype):
    v = _f32(v)
    out = np.empty_like(v)
    lib().msq_oracle_round_lowp(_p(v), _p(out), C.c_int64(v.size), C.c_int(LOWP[str(dtype).replace("torch.", "")]))
    return out


def quantize_mx(a, scale_bits, elem_fmt, axis=-1, block_size=0, round="nearest",
                flush_fp32_subnorms=False, plus_eps_defect=False):
    a = _f32(a)
    pre, axis_len, post = _pap(a.shape, axis)
    out = np.empty_like(a)
    st = lib().msq_oracle_quantize_mx(_p(a), _p(out), C.c_int64(pre), C.c_int64(axis_len),
                                      C.c_int64(post), C.c_int(block_size), elem_fmt.lower().encode(),
                                      C.c_int(scale_bits), C.c_int(RD[round]),
                                      C.c_int(bool(flush_fp32_subnorms)), C.c_int(bool(plus_eps_defect)))
    if st < 0:
        raise Exception("Undefined elem format", elem_fmt)
    return out


def quantize_mx_lowp(a, dtype, scale_bits, elem_fmt, axis=-1, block_size=0, round="nearest", flush_fp32_subnorms=False):
    """mx_ops.py:332-457 (_quantize_mx, Python path) computed IN the tensor dtype (fp16 / bf16): `a` holds the tensor's values
    as float32."""
    a = _f32(a)
    pre, axis_len, post = _pap(a.shape, axis)
    out = np.empty_like(a)
    st = lib().msq_oracle_quantize_mx_lowp(_p(a), _p(out), C.c_int(LOWP[str(dtype).replace("torch.", "")]), C.c_int64(pre),
                                           C.c_int64(axis_len), C.c_int64(post), C.c_int(block_size), elem_fmt.lower().encode(),
                                           C.c_int(scale_bits), C.c_int(RD[round]), C.c_int(bool(flush_fp32_subnorms)))
    if st < 0:
        raise Exception("Undefined elem format / unsupported dtype", elem_fmt, dtype)
    return out


def quantize_mx_native(a, scale_bits, ebits, mbits, max_norm, tile, axis, flush=False, rmode=0):
    a = _f32(a)
    pre, axis_len, post = _pap(a.shape, axis)
    out = np.empty_like(a)
    lib().msq_oracle_quantize_mx_native(_p(a), _p(out), C.c_int64(pre), C.c_int64(axis_len),
                                        C.c_int64(post), C.c_int(tile), C.c_int(scale_bits),
                                        C.c_int(ebits), C.c_int(mbits), C.c_float(max_norm),
                                        C.c_int(bool(flush)), C.c_int(rmode))
    return out


def reduce_inner(a, is_max):
    a = _f32(a)
    inner = a.shape[-1]
    outer = a.size // inner
    out = np.empty(a.shape[:-1], dtype=np.float32)
    lib().msq_oracle_reduce_inner(_p(a), _p(out), C.c_int64(outer), C.c_int64(inner), C.c_int(is_max))
    return out


def linear(x, w, bias=None):
    x, w = _f32(x), _f32(w)
    M, K = x.shape
    N = w.shape[0]
    b = _f32(bias) if bias is not None else None
    y = np.empty((M, N), dtype=np.float32)
    lib().msq_oracle_linear(_p(x), _p(w), _p(b), _p(y), C.c_int64(M), C.c_int64(N), C.c_int64(K))
    return y


def posit_decode(code, n, es):
    return lib().msq_oracle_posit_decode(int(code), n, es)


def posit_encode(v, n, es):
    return lib().msq_oracle_posit_encode(float(v), n, es)


def kv_group_quant(x, quantize_bit, group_size, along_tokens, dtype="float32"):
    """GEAR group fake-quant of a [B, H, S, D] cache tensor (values held as float32; `dtype` = the tensor's own dtype,
    the result is cast back to it): along_tokens=False -> fake_groupwise_token_asymmetric_quantization
    (compress_function.py:8-38), True -> fake_groupwise_channel_asymmetric_quantization_new (:41-70)."""
    x = _f32(x)
    B, H, S, D = x.shape
    out = np.empty_like(x)
    dt = {"float32": 0, "f32": 0}.get(str(dtype).replace("torch.", ""), None)
    if dt is None:
        dt = LOWP[str(dtype).replace("torch.", "")]
    rc = lib().msq_oracle_kv_group_quant(_p(x), _p(out), C.c_int(dt), C.c_int64(B), C.c_int64(H), C.c_int64(S), C.c_int64(D),
                                         C.c_int(quantize_bit), C.c_int64(group_size), C.c_int(bool(along_tokens)))
    if rc:
        raise ValueError("group_size should be a factor of the grouped dimension size")
    return out


def _vq(bits, exp_bits, max_norm, round, allow_denorm):
    return (C.c_int(bits), C.c_int(exp_bits), C.c_float(max_norm), C.c_int(RD[round]), C.c_int(bool(allow_denorm)))


def vec_layernorm(x, w, b, eps, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    """mx LayerNorm forward (layernorm.py:18-42 / norm_utils.py:27-113) with every op rounded by Q = (bits, exp_bits)."""
    x = _f32(x); w = _f32(w); b = _f32(b)
    H = x.shape[-1]
    out = np.empty_like(x)
    lib().msq_oracle_vec_layernorm(_p(x), _p(w), _p(b), _p(out), C.c_int64(x.size // H), C.c_int64(H), C.c_double(eps),
                                   *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out


def vec_gelu(x, first_order=False, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    x = _f32(x)
    out = np.empty_like(x)
    lib().msq_oracle_vec_gelu(_p(x), _p(out), C.c_int64(x.size), C.c_int(bool(first_order)),
                              *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out


def vec_rmsnorm(x, w, b, eps, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    """mx RMSNorm forward (layernorm.py:98-128) with every op rounded by Q = (bits, exp_bits)."""
    x = _f32(x); w = _f32(w); b = _f32(b)
    H = x.shape[-1]
    out = np.empty_like(x)
    lib().msq_oracle_vec_rmsnorm(_p(x), _p(w), _p(b), _p(out), C.c_int64(x.size // H), C.c_int64(H), C.c_double(eps),
                                 *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out


def vec_silu(x, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    """mx silu forward (activations.py:420-434)"""
    x = _f32(x)
    out = np.empty_like(x)
    lib().msq_oracle_vec_silu(_p(x), _p(out), C.c_int64(x.size), *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out


def vec_mul(a, b, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    """simd_mul of two tensors of one shape (simd_ops.py:154-187)"""
    a = _f32(a); b = _f32(b)
    out = np.empty_like(a)
    lib().msq_oracle_vec_mul(_p(a), _p(b), _p(out), C.c_int64(a.size), *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out


def vec_add(a, b, bits=9, exp_bits=8, max_norm=3.3895313892515355e38, round="nearest", allow_denorm=True):
    a = _f32(a); b = _f32(b)
    out = np.empty_like(a)
    lib().msq_oracle_vec_add(_p(a), _p(b), _p(out), C.c_int64(a.size), *_vq(bits, exp_bits, max_norm, round, allow_denorm))
    return out
