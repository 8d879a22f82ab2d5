"""Round-2 GPU tests: advisor findings (dtype of the MX-path modules inside an fp16 model, MXLinear.pack() on 3-D
inputs, argument guards of the C ABI) and the new hot-path pieces of this round."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def msq():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import msq as m
    m._lib.lib()
    return m


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def dev():
    return torch.device("cuda:0")


def _tiny_llama(dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).eval().to(dtype).to(dev())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("path", ["mx", "bf16"])
def test_pack_layers_in_half_precision_model(msq, dtype, path):
    """Llama-2 / OPT checkpoints load as fp16 (get_llama, torch_dtype='auto'): every packed projection must hand the
    model's dtype back, otherwise the first dense Linear behind it fails with a dtype mismatch (advisor, round 1)."""
    from msq.harness import find_layers
    from msq.harness.data_utils import _Enc
    from msq.harness.evalppl import pack_layers, perplexity, quantize_layers_nearest
    tokens = _Enc(torch.randint(0, 512, (1, 64 * 4), generator=torch.Generator().manual_seed(1)))
    m = _tiny_llama(dtype)
    quantize_layers_nearest(m.model.layers, dev(), dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3",
                                                        axes=[-1], block_size=32))
    ppl_dense = perplexity(m, tokens, dev(), 64)
    packed, dense = pack_layers(m.model.layers, path=path)
    assert packed == 14 and dense == 0
    kinds = {type(l).__name__ for layer in m.model.layers for l in layer.modules()}
    assert ("MXLinearW4A8" in kinds) == (path == "mx")
    x = torch.randn(2, 5, 256, device=dev(), dtype=dtype)
    y = m.model.layers[0].self_attn.q_proj(x)
    assert y.dtype == dtype and y.shape == (2, 5, 256)
    ppl = perplexity(m, tokens, dev(), 64)                         # runs end to end: no dtype mismatch at lm_head / SDPA
    # bf16-activation path: same weights, bf16 rounding of activations only; MX path adds MX-FP8 activations
    assert abs(ppl - ppl_dense) / ppl_dense < (0.02 if path == "bf16" else 0.05), (ppl, ppl_dense)


def test_mxlinear_pack_3d_input_matches_unpacked(msq):
    """MXLinear quantises activations along axes=[1] (number_system/mx/linear.py:66-73): for [B, S, K] that is the
    SEQUENCE axis.  pack() must keep that: same outlier masks / scales as the unpacked module (advisor, round 1)."""
    sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8,
                                      "block_size": 32, "bfloat": 16, "custom_cuda": True})
    torch.manual_seed(3)
    lin = msq.linear.MXLinear(128, 256, True, mx_specs=sp).to(dev())
    X = torch.randn(3, 64, 128, device=dev())
    X[torch.rand(3, 64, 128, device=dev()) < 0.02] *= 10
    with torch.no_grad():
        y_ref = lin(X)
        # the activation operand the unpacked forward builds (blocks of 32 along S)
        bf_in = msq.elemwise_ops.quantize_elemwise_op(X, mx_specs=lin.mx_specs, round=lin.mx_specs["round_output"])
        q_seq = msq.mx_ops.quantize_mx_outlier_op(bf_in, lin.mx_specs, inlier_elem_format="fp8_e4m3",
                                                  outlier_elem_format="fp8_e4m3", axes=[1], round=lin.mx_specs["round_mx_output"])
        q_feat = msq.mx_ops.quantize_mx_outlier_op(bf_in, lin.mx_specs, inlier_elem_format="fp8_e4m3",
                                                   outlier_elem_format="fp8_e4m3", axes=[2], round=lin.mx_specs["round_mx_output"])
        assert not torch.equal(q_seq, q_feat)                      # the two groupings really differ on this input
        y_packed = lin.pack()(X)
        y2 = lin(X[0])                                             # 2-D input: axis 1 is the feature axis, fused path
    assert y_packed.shape == y_ref.shape == (3, 64, 256)
    err = (y_packed - y_ref).abs()
    tol = y_ref.abs() * 2.0 ** -7 + 1e-6                           # one bf16 ulp of the re-rounded output (fp32 sum order)
    assert bool((err <= tol).all()), float(err.max())
    assert float((err > 0).float().mean()) <= 0.01
    lin._packed = None
    with torch.no_grad():
        y2_ref = lin(X[0])
    e2 = (y2 - y2_ref).abs()
    assert bool((e2 <= y2_ref.abs() * 2.0 ** -7 + 1e-6).all())


def test_qlinear_rejects_over_4gib_operands(msq):
    """msq_qlinear_bf16 addresses activations and packed planes with 32-bit buffer offsets: operands above 4 GiB must be
    refused with MSQ_ERR_UNSUPPORTED, not wrap around silently (advisor, round 1).  No memory is touched: the check
    happens before any launch, so the pointers may be small dummies."""
    L = msq._lib.lib()
    d = torch.zeros(1024, dtype=torch.uint8, device=dev())
    p = msq._lib.ptr
    # M * K * 2 bytes > 4 GiB
    rc = L.msq_qlinear_bf16(p(d), None, p(d), p(d), None, p(d), 2, 1 << 20, 256, 4096, 32, 0, 5, None, 0, None)
    assert rc == -2, rc
    assert b"4 GiB" in L.msq_last_error()
    # a packed plane > 4 GiB: N * K bytes for the unified layout
    rc = L.msq_qlinear_bf16(p(d), None, p(d), p(d), None, p(d), 2, 16, 1 << 17, 1 << 16, 32, 0, 5, None, 0, None)
    assert rc == -2, rc


def test_decode_kernels_on_second_stream_and_repeated(msq):
    """The single-launch decode kernels need 120 KiB of dynamic LDS: the attribute is set per device inside the
    library (no process-wide flag).  Repeated calls from two streams must keep returning identical results."""
    g = torch.Generator(device=dev()).manual_seed(5)
    W = torch.randn(8192, 4096, generator=g, device=dev()) * 0.02
    P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
    x = torch.randn(4, 4096, generator=g, device=dev()).to(torch.bfloat16)
    y0 = msq.qlinear.qlinear(x, P, None, torch.float32)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        y1 = msq.qlinear.qlinear(x, P, None, torch.float32)
    s.synchronize()
    assert torch.equal(y0, y1)
    ref = x.float() @ msq.qlinear.unpack_weight(P).t()
    assert float((y0 - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


# ---------------------------------------------------------------- fp16 / bf16 RTN path: compute in the tensor dtype
def _bits_to_torch(u16, dn):
    t = torch.from_numpy(u16.view(np.int16).copy())
    return t.view(torch.float16 if dn == "f16" else torch.bfloat16)


def test_lowp_floor_log2_exhaustive_gpu(msq):
    """The device rule for floor(log2(x)) on half tensors against torch's CPU result for EVERY positive finite fp16 and
    bf16 value (fixture generated by tests/golden/make_golden_lowp.py)."""
    z = np.load(os.path.join(G, "log2_lowp.npz"))
    L = msq._lib.lib()
    for dn, code, top in (("f16", 1, 0x7C00), ("bf16", 2, 0x7F80)):
        x = _bits_to_torch(np.arange(1, top, dtype=np.uint16), dn).float().to(dev())
        out = torch.empty_like(x)
        msq._lib.check(L.msq_floor_log2_lowp(msq._lib.ptr(x), msq._lib.ptr(out), x.numel(), code,
                                             msq._lib.current_stream(dev())), "msq_floor_log2_lowp")
        assert (out.cpu().numpy() == z[dn].astype(np.float32)).all(), dn
    zero = torch.zeros(4, device=dev()); o = torch.empty(4, device=dev())
    msq._lib.check(L.msq_floor_log2_lowp(msq._lib.ptr(zero), msq._lib.ptr(o), 4, 1, msq._lib.current_stream(dev())), "log2")
    assert bool(torch.isinf(o).all()) and bool((o < 0).all())


def test_lowp_outlier_fakequant_golden_gpu(msq):
    """quantize_mx_outlier_v1 on fp16 / bf16 tensors against the reference run on the same half tensors (264 cases:
    harness default int2 / fp4 along out_features, the BASELINE formats, three rounding modes, scale bits 4, blocks
    8 ... 128, values just under powers of two, fp16 subnormals): values AND outlier masks bit for bit, AssertionError
    exactly where the reference's NaN assertion fired."""
    z = np.load(os.path.join(G, "outlier_lowp.npz"))
    meta = json.load(open(os.path.join(G, "outlier_lowp_meta.json")))
    n = na = 0
    for key, m in sorted(meta.items()):
        dn, tname, cname = key.split("|")
        isb, osb, fi, fo, sd, axes, bs, rnd = m["cfg"]
        A = _bits_to_torch(z[f"in|{dn}|{tname}"], dn).to(dev())
        if "assert" in m:
            with pytest.raises(AssertionError):
                msq.quant.quantize_mx_outlier_v1(A, isb, osb, fi, fo, "max", sd, axes, bs, rnd)
            na += 1
            continue
        r = msq.quant.outlier_fakequant(A, isb, osb, fi, fo, sd, axes[0], bs, rnd, want_mask=True)
        assert r["out"].dtype == A.dtype
        got = r["out"].view(torch.int16).cpu().numpy().view(np.uint16)
        ref = z[f"out|{key}"]
        nan_both = np.isnan(_bits_to_torch(got, dn).float().numpy()) & np.isnan(_bits_to_torch(ref, dn).float().numpy())
        assert ((got == ref) | nan_both).all(), (key, int(((got != ref) & ~nan_both).sum()))
        mask = np.unpackbits(z[f"mask|{key}"])[:A.numel()].reshape(tuple(A.shape))
        assert (r["mask"].cpu().numpy() == mask).all(), (key, "mask")
        y = msq.quant.quantize_mx_outlier_v1(A, isb, osb, fi, fo, "max", sd, axes, bs, rnd)     # reference-named entry
        assert torch.equal(y.view(torch.int16), r["out"].view(torch.int16))
        n += 1
    assert n >= 230 and na >= 20


@pytest.mark.parametrize("dn", ["f16", "bf16"])
def test_lowp_llama_sized_weight_vs_oracle(msq, O, dn):
    """A Llama-2-7B sized projection [4096 x 4096] in the checkpoint dtype through the harness call
    (llm/llama.py:229-253: int2 / fp4, blocks of 16 along out_features) and the BASELINE format pair: HIP == oracle
    (compute-in-dtype restatement, pinned on the reference's half-tensor goldens) on all 16.8 M weights."""
    dt = torch.float16 if dn == "f16" else torch.bfloat16
    g = torch.Generator(device=dev()).manual_seed(21)
    W = (torch.randn(4096, 4096, generator=g, device=dev()) * 0.02)
    W[torch.rand(4096, 4096, generator=g, device=dev()) < 0.005] *= 16
    W = W.to(dt)
    Wf = W.float().cpu().numpy()
    for fi, fo, ax, bs in (("int2", "fp4", 0, 16), ("fp4_e2m1", "fp8_e4m3", -1, 32)):
        o = O.outlier_fakequant_lowp(Wf, dn, 8, 8, fi, fo, 2, ax, bs)
        # In fp16 the harness default can hit the reference's own NaN assertion on a handful of blocks (all inliers
        # zero -> e_in = -20 -> outliers * 2^-20 -> e_out = -26 -> 2^-26 underflows to 0 in fp16 -> x / 0): the
        # oracle reports it (status 1) and the shim must raise exactly then, like utils/quant.py:225-250.
        if o["status"] & 1:
            with pytest.raises(AssertionError):
                msq.quant.outlier_fakequant(W, 8, 8, fi, fo, 2, ax, bs)
        keep = msq.quant.CHECK_NAN
        msq.quant.CHECK_NAN = False                                  # compare the tensors themselves, NaNs included
        try:
            r = msq.quant.outlier_fakequant(W, 8, 8, fi, fo, 2, ax, bs, want_mask=True)
        finally:
            msq.quant.CHECK_NAN = keep
        got = r["out"].float().cpu().numpy()
        assert (r["mask"].cpu().numpy() == o["mask"]).all()
        assert ((got == o["out"]) | (np.isnan(got) & np.isnan(o["out"]))).all()
        assert int(np.isnan(got).sum()) == int(np.isnan(o["out"]).sum())
        # and it is NOT what "upcast, compute in fp32, round once" gives
        r32 = msq.quant.outlier_fakequant(W, 8, 8, fi, fo, 2, ax, bs, compute_dtype="float32")["out"]
        frac = float(((r32 != r["out"]) & ~torch.isnan(r["out"])).float().mean())
        assert 0 < frac < 0.05, frac
