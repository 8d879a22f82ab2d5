"""Round-5 GPU tests: every fused-GEMM kernel FORCED and compared with the oracle on multi-round grids (judge, round 4, weak 2), the
persistent stream-K kernel k_qgemm256p (cut tiles summed through workspace slots), the register-exchange epilogue."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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


def _weights(N, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, K, generator=g) * 0.02
    W[torch.rand(N, K, generator=g) < 0.01] *= 16
    return W


def _plan(msq, M, N, K, cus=0):
    v = [ctypes.c_int(0) for _ in range(4)]
    ws = ctypes.c_int64(0)
    rc = msq._lib.lib().msq_qgemm256p_plan(M, N, K, cus, *[ctypes.byref(x) for x in v], ctypes.byref(ws))
    return rc, [x.value for x in v], ws.value


# ----------------------------------------------------------------------------------------------------------------------
# forced kernels against the oracle (O.linear on the oracle's own fake-quant weight), multi-round grids
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(4352, 4096, 256), (2048 + 19, 8704, 128)])
@pytest.mark.parametrize("fo", ["posit8_es1", "fp8_e4m3"])
def test_forced_gemm_kernels_against_the_oracle_on_multi_round_grids(msq, O, M, N, K, fo, monkeypatch):
    """k_qgemm256<MF 16> (MSQ_GEMM_256=1), its 128-row form (=2), the persistent kernel (=3) and k_qgemm3 (=0), each FORCED, on grids of
    more than 256 blocks (17 x 16 = 272 and 9 x 34 = 306 tiles of 256 x 256: two rounds over the 256 CUs, a ragged last row tile, a panel
    count that is not a multiple of 8) against O.linear(X, W_oracle) -- the oracle's restatement of F.linear
    (number_system/mx/linear.py:91) on the ORACLE's fake-quant weight (utils/quant.py:147-266), not on anything the GPU unpacked:
    within 2e-5 max|y| (fp32 accumulation against double), identical over 30 launches."""
    W = _weights(N, K, 21)
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(22)).to(torch.bfloat16)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(23))
    Wo = O.outlier_fakequant(W.numpy(), 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"]
    ref = O.linear(X.float().numpy(), Wo, bias.numpy())
    P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    Xd, bd = X.to(dev()), bias.to(dev())
    tol = 2e-5 * np.abs(ref).max() + 1e-6
    for flag in ("0", "1", "2", "3"):
        monkeypatch.setenv("MSQ_GEMM_256", flag)
        y = msq.qlinear.qlinear(Xd, P, bd, torch.float32)
        assert np.abs(y.cpu().numpy() - ref).max() <= tol, flag
        for _ in range(30):
            assert torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.float32), y), flag
    monkeypatch.delenv("MSQ_GEMM_256")


# ----------------------------------------------------------------------------------------------------------------------
# k_qgemm256p: persistent blocks, stream-K over the part-filled last round
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(130, 512, 1024), (256, 256, 8192), (600, 768, 2048), (1000, 2304, 640), (2048, 4096, 4096), (513, 11008, 1024)])
@pytest.mark.parametrize("fo", ["posit8_es1", "fp8_e4m3"])
def test_persistent_stream_k_cut_tiles(msq, O, M, N, K, fo, monkeypatch):
    """Shapes whose plan cuts tiles along K (msq_qgemm256p_plan: ws_bytes > 0): 2 tiles cut in 2, ONE tile cut 16 ways, 9 tiles cut 4 ways,
    36 tiles over 45 blocks, the true Llama-2-7B o_proj (128 tiles cut in 2 over 256 blocks), 129 tiles with a panel count that is not a
    multiple of 8.  A cut tile is the fp32 sum of its pieces' accumulators in K order (fixed order: the launches repeat bit for bit, also
    when the workspace is reused with stale flags from the previous launch -- the launcher zeroes them); against the oracle within 2e-5
    max|y|, with a bias, float32 and bfloat16 outputs, ragged M."""
    rc, (Pb, full, R, q), wsb = _plan(msq, M, N, K)
    assert rc == 0 and wsb > 0 and q < K // 64
    W = _weights(N, K, 31)
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(32)).to(torch.bfloat16)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(33))
    Wo = O.outlier_fakequant(W.numpy(), 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"]
    ref = O.linear(X.float().numpy(), Wo, bias.numpy())
    P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    Xd, bd = X.to(dev()), bias.to(dev())
    monkeypatch.setenv("MSQ_GEMM_256", "3")
    assert msq._lib.lib().msq_qlinear_workspace_bytes(M, N, K) >= wsb
    y = msq.qlinear.qlinear(Xd, P, bd, torch.float32)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    yb = msq.qlinear.qlinear(Xd, P, bd, torch.bfloat16)
    assert torch.equal(yb, y.to(torch.bfloat16))                   # the same sums, rounded once
    for _ in range(20):
        assert torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.float32), y)
        junk = torch.randn(1 << 20, device=dev())                  # churn the allocator: the next workspace holds other bytes
        del junk
    monkeypatch.setenv("MSQ_GEMM_256", "1")
    a = msq.qlinear.qlinear(Xd, P, bd, torch.float32)               # the uncut sum (k_qgemm256): another fp32 order, within rounding
    assert (a - y).abs().max().item() <= 2e-5 * np.abs(ref).max()
    monkeypatch.delenv("MSQ_GEMM_256")


@pytest.mark.parametrize("M,N,K", [(2048, 16384, 256), (4096 - 37, 4096, 384), (777, 16384, 1152)])
def test_persistent_uncut_tiles_equal_qgemm256_bit_for_bit(msq, M, N, K, monkeypatch):
    """Whole rounds of tiles (no cut): every output element accumulates the same products in the same order as in k_qgemm256 -- equal
    bits, with several tiles per block (the stores of one tile in flight under the first K-step of the next), a bias, ragged M (rows
    beyond M are dropped by the range check of the tile's buffer descriptor), three output dtypes."""
    rc, (Pb, full, R, q), wsb = _plan(msq, M, N, K)
    assert rc == 0 and wsb == 0
    W = _weights(N, K, 11).to(dev())                              # (seed of the round-4 test: every group fits the unified layout)
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(42)).to(dev()).to(torch.bfloat16)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(43)).to(dev())
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            for b in (None, bias):
                monkeypatch.setenv("MSQ_GEMM_256", "1")
                a = msq.qlinear.qlinear(X, P, b, dt)
                monkeypatch.setenv("MSQ_GEMM_256", "3")
                for _ in range(4):
                    assert torch.equal(msq.qlinear.qlinear(X, P, b, dt), a), (fo, dt)
    monkeypatch.delenv("MSQ_GEMM_256")


def test_persistent_kernel_without_workspace_falls_back(msq, monkeypatch):
    """msq_qlinear_bf16 with a NULL workspace on a shape whose plan cuts tiles: the persistent kernel is not taken (the C ABI promises
    'NULL or too small = single pass, never an error'), the result is that of the other kernels."""
    M, N, K = 600, 768, 2048
    W = _weights(N, K, 51).to(dev())
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(52)).to(dev()).to(torch.bfloat16)
    P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
    L, ptr = msq._lib.lib(), msq._lib.ptr
    y = torch.empty(M, N, device=dev())
    monkeypatch.setenv("MSQ_GEMM_256", "3")
    rc = L.msq_qlinear_bf16(ptr(X), ptr(P.inl), ptr(P.out), ptr(P.scl), None, ptr(y), 0, M, N, K, P.block, P.in_kind, P.out_kind, None, 0,
                            msq._lib.current_stream(dev()))
    assert rc == 0
    monkeypatch.setenv("MSQ_GEMM_256", "0")
    ref = msq.qlinear.qlinear(X, P, None, torch.float32)
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    monkeypatch.delenv("MSQ_GEMM_256")


# ----------------------------------------------------------------------------------------------------------------------
# plain-MX surfaces just under powers of two (advisor, round 4): Python-path surfaces follow floor(torch.log2), native ones the exponent field
# ----------------------------------------------------------------------------------------------------------------------
def _planted(rows, K, seed, lo=-12, hi=12):
    """[rows, K] float32 whose blocks of 32 each hold ONE maximum 1 .. 90 ulps under a power of two 2^u (u in lo .. hi, either sign) and
    30 smaller values: floor(torch.log2(max)) is u for the nearest of them (up to 88 ulps for |u| >= 128, fewer near 1: msq_device.h
    ilog2f_torch), u - 1 beyond -- the block scale of the reference's Python path doubles exactly there."""
    rs = np.random.RandomState(seed)
    x = rs.randn(rows, K).astype(np.float32)
    nb = K // 32
    for r in range(rows):
        for b in range(nb):
            u = int(rs.randint(lo, hi + 1))
            d = int(rs.randint(1, 91)) if rs.rand() < 0.3 else int(rs.randint(1, 7))     # (K = 1 .. 11 for |u| <= 31: mostly bumped, some not)
            top = np.float32(2.0 ** u).view(np.uint32) - np.uint32(d)
            top = np.uint32(top).view(np.float32)
            blk = x[r, b * 32:(b + 1) * 32]
            blk *= np.float32(0.45 * 2.0 ** u / max(np.abs(blk).max(), 1e-30))
            blk[rs.randint(0, 32)] = top if rs.rand() < 0.5 else -top
    return x


def _eq(a, b):
    """equal values (a zero of either sign is the same value: the native bit codec returns +0 where the reference's `sign * 0` gives -0)"""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def _e4m3_decode(e):
    e = e.astype(np.int32)
    sgn = np.where(e & 0x80, -1.0, 1.0); ex = (e >> 3) & 15; mant = e & 7
    return sgn * np.where(ex > 0, (1 + mant / 8.0) * np.exp2(ex - 7.0), mant / 8.0 * 2.0 ** -6)


@pytest.mark.parametrize("fmt", ["fp8_e4m3", "fp4_e2m1", "fp6_e3m2", "int8"])
@pytest.mark.parametrize("rnd", ["nearest", "floor", "even"])
def test_quantize_mx_python_path_just_below_powers_of_two(msq, O, fmt, rnd):
    """`_quantize_mx` of the package (= the reference's Python path, mx_ops.py:332-457: shared exponent floor(torch.log2(max)) in fp32, under
    `floor` the private exponents too) against the oracle on blocks whose maximum sits 1 .. 90 ulps under a power of two, along the inner
    and along an outer axis (both kernel families), fp32; the NATIVE entry (quantize_mx_by_tile_func_cuda, cpp/shared_exp.cuh: the
    exponent field) against the oracle's native restatement on the same data -- and the two must DIFFER here (the planted data bite)."""
    A = _planted(24, 256, 7)
    At = torch.from_numpy(A).to(dev())
    y = msq.mx_ops._quantize_mx(At, 8, fmt, axes=[-1], block_size=32, round=rnd).cpu().numpy()
    yo = O.quantize_mx(A, 8, fmt, axis=-1, block_size=32, round=rnd)
    assert _eq(y, yo)
    B = np.ascontiguousarray(A.reshape(24, 8, 32).transpose(0, 2, 1))            # [24, 32, 8]: blocks of 32 along axis 1 (outer-axis kernels)
    Bt = torch.from_numpy(B).to(dev())
    yb = msq.mx_ops._quantize_mx(Bt, 8, fmt, axes=[1], block_size=32, round=rnd).cpu().numpy()
    assert _eq(yb, O.quantize_mx(B, 8, fmt, axis=1, block_size=32, round=rnd))
    e, m, ex, mx, mn = msq.formats._get_format_params(fmt)
    rm = {"nearest": 0, "floor": 1, "even": 2}[rnd]
    yn = msq.funcs.quantize_mx_by_tile_func_cuda(At, 8, e, m, mx, 32, 1, False, rm).cpu().numpy()
    assert _eq(yn, O.quantize_mx_native(A, 8, e, m, mx, 32, 1, False, rm))
    assert (yn != y).any()                                                       # the doubled block scales show (where an element needed the lost bit)


def test_mx_operand_packers_and_kv_just_below_powers_of_two(msq, O):
    """The MX operand packers (activations: e4m3 codes + scale bytes, both kernels; weights: fp4 / fp6 / e4m3-value planes) and the KV-cache MX
    quantiser stand in for the reference's CPU `_quantize_mx` (custom_cuda = False): on planted blocks they decode to the oracle's
    Python-path values bit for bit -- scale bytes included."""
    X = _planted(40, 512, 9, -8, 8)
    Xo = O.quantize_mx(X, 8, "fp8_e4m3", axis=-1, block_size=32)
    for src in (torch.from_numpy(X).to(dev()),):
        xc, xs = msq.qlinear.mx_pack_act(src, check_status=True)
        dec = _e4m3_decode(xc.cpu().numpy()) * np.repeat(np.exp2(xs.cpu().numpy().astype(np.float64) - 127.0), 32, axis=1)
        assert (dec == Xo.astype(np.float64)).all()
    # bf16 activations: every bf16 is an fp32 value; a bf16 has 8 significand bits, so "just under a power of two" = its top code
    Xb = torch.from_numpy(X).to(dev()).to(torch.bfloat16)
    xc, xs = msq.qlinear.mx_pack_act(Xb, check_status=True)
    dec = _e4m3_decode(xc.cpu().numpy()) * np.repeat(np.exp2(xs.cpu().numpy().astype(np.float64) - 127.0), 32, axis=1)
    assert (dec == O.quantize_mx(Xb.float().cpu().numpy(), 8, "fp8_e4m3", axis=-1, block_size=32).astype(np.float64)).all()
    # weights through the GEMM itself: x = identity rows pick the decoded weight columns back out (exact: one product per output)
    W = _planted(256, 256, 11, -6, 6)
    eye = torch.eye(256, device=dev())
    for w_fmt, ofmt in (("e2m1", "fp4_e2m1"), ("e3m2", "fp6_e3m2")):
        P = msq.qlinear.mx_pack_weight(torch.from_numpy(W).to(dev()), w_fmt=w_fmt)
        Wd = msq.qlinear.qlinear_mx_w4a8(eye, P, None, torch.float32).t().cpu().numpy()      # y[i, n] = W_q[n, i]
        Wo = O.quantize_mx(W, 8, ofmt, axis=-1, block_size=32)
        assert np.array_equal(Wd, Wo), w_fmt
    # KV cache, fp32: keys in blocks along the tokens, values along head_dim
    k = torch.from_numpy(_planted(2 * 4 * 64, 64, 13, -6, 6).reshape(2, 4, 64, 64)).to(dev())
    kq = msq.kvcache.mx_quantize_values(k, "fp8_e4m3", 32)                       # blocks along head_dim = the planted axis
    assert np.array_equal(kq.cpu().numpy(), O.quantize_mx(k.cpu().numpy(), 8, "fp8_e4m3", axis=3, block_size=32))
    kt = k.transpose(2, 3).contiguous()                                          # planted axis -> tokens
    kq = msq.kvcache.mx_quantize_keys(kt, "fp4_e2m1", 32)
    assert np.array_equal(kq.cpu().numpy(), O.quantize_mx(kt.cpu().numpy(), 8, "fp4_e2m1", axis=2, block_size=32))


# ----------------------------------------------------------------------------------------------------------------------
# defaults match the reference (judge, round 4, next 8)
# ----------------------------------------------------------------------------------------------------------------------
def test_gptq_defaults_are_the_references(msq):
    """harness/gptq.py: FACTOR_FP64 is False by default -- the inverse-Hessian factor is torch's float32 Cholesky like llm/gptq.py:98-104, and
    a Hessian that is not positive-definite IN FLOAT32 raises there, as the reference does: two identical input columns and a damping of
    1e-9 of the mean diagonal (1 + 1e-9 == 1 in float32: the damped matrix is exactly rank-deficient).  With the reference's default
    damping (percdamp = 0.01) the same Hessian factors and the solve runs."""
    import msq.harness.gptq as Gm
    assert Gm.FACTOR_FP64 is False and Gm.UPDATE_FP64 is False

    def run(percdamp):
        lin = torch.nn.Linear(64, 32, bias=False).to(dev())
        gp = Gm.GPTQ(lin)
        gp.quantizer = msq.quant.MXQuantizer()
        gp.quantizer.configure(8, 8, "fp4_e2m1", "fp8_e4m3", axes=[0], block_size=16)
        H = torch.eye(64, device=dev())
        H[0, 1] = H[1, 0] = 1.0                                      # columns 0 and 1 of the inputs are the same signal
        gp.H = H
        gp.nsamples = 1
        gp.fasterquant(blocksize=32, percdamp=percdamp, verbose=False)
        return lin.weight

    with pytest.raises(Exception) as ei:
        run(1e-9)
    assert "positive-definite" in str(ei.value) or "positive definite" in str(ei.value), str(ei.value)
    assert torch.isfinite(run(0.01)).all()


# ----------------------------------------------------------------------------------------------------------------------
# perplexity fixture, round 5: the harness's own default quantiser (int2 / fp4, axes = [0], block 16) as a second leg
# ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ppl_default_leg(msq):
    import bench
    for k in ("MSQ_PPL_MODEL", "MSQ_WIKITEXT2_DIR", "MSQ_PPL_SEQLEN", "MSQ_PPL_NSAMPLES", "MSQ_PPL_DISABLE"):
        os.environ.pop(k, None)
    return bench.ppl_delta_from_env(dev(), "fp4_e2m1", "posit8_es1", 32, paths=("harness_default",))


def test_ppl_harness_default_config_leg(ppl_default_leg):
    """The reference harness's OWN quantiser configuration (llm/llama.py:229-237: int2 inliers, fp4 outliers, blocks of 16 along
    out_features) on the round-5 fixture (a gazetteer corpus: most tokens are entity names, digits and world facts a model holds in its
    weights): here the quantiser is NOT near-lossless -- the CPU reference's perplexity sits >= 1 % above the unquantised model's -- and the
    HIP path (in-dtype quantiser -> values packed as they are -> fused GEMM) stays within 0.05 of it, KL <= 1e-3 nats / token, top-1 >= 98 %."""
    r = ppl_default_leg
    h = r["harness_default"]
    assert h["layers_kept_dense"] == 0 and h["layers_packed"] == 28 and r["windows"] >= 40
    assert h["ppl_cpu_reference"] >= 1.01 * r["ppl_unquantised_cpu_fp32"], (h["ppl_cpu_reference"], r["ppl_unquantised_cpu_fp32"])
    assert abs(h["delta"]) <= 0.05 and abs(h["relative_delta"]) < 0.05 / 5.5, h
    lm = h["logits_vs_cpu_reference"]
    assert lm["mean_kl_nats_per_token"] <= 1e-3 and lm["top1_agreement"] >= 0.98, lm


def test_ppl_fixture_shows_one_flipped_code_bit(msq, ppl_default_leg):
    """ONE bit of ONE packed code of ONE layer (the top bit of an e4m3 code in the value plane: that weight changes sign -- what a wrong
    outlier-mask bit does to an element, its value taken from the other quantiser) changes the logits of the packed model measurably
    (largest logit error and KL against the CPU reference both move); 64 flipped codes raise the KL and move the perplexity delta."""
    import bench

    def corrupt(n):
        def f(model):
            q = model.model.layers[2].mlp.down_proj
            cp = q.out_plane
            assert cp.numel() > 4096 and cp.dtype == torch.uint8
            for i in range(n):
                j = 2048 + 16 * i
                cp[j] = int(cp[j].item()) ^ 0x80
        return f
    clean = ppl_default_leg["harness_default"]
    one = bench.ppl_delta_from_env(dev(), "fp4_e2m1", "posit8_es1", 32, paths=("harness_default",), corrupt=corrupt(1))["harness_default"]
    assert one["logits_vs_cpu_reference"]["max_logit_abs_err"] != clean["logits_vs_cpu_reference"]["max_logit_abs_err"]
    assert one["logits_vs_cpu_reference"]["mean_kl_nats_per_token"] != clean["logits_vs_cpu_reference"]["mean_kl_nats_per_token"]
    many = bench.ppl_delta_from_env(dev(), "fp4_e2m1", "posit8_es1", 32, paths=("harness_default",), corrupt=corrupt(64))["harness_default"]
    assert many["logits_vs_cpu_reference"]["mean_kl_nats_per_token"] > clean["logits_vs_cpu_reference"]["mean_kl_nats_per_token"]
    assert many["delta"] != clean["delta"]


def _eq_bits(a, b):
    """the same float32 bit patterns (the sign of a zero included), NaN where NaN"""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("FMT", ["fp8_e4m3", "fp4_e2m1"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kv_mx_e4m3_block_setup_every_binade(msq, O, dtype, FMT):
    """The e4m3 convert path of the KV-cache MX quantiser decides per block, from the T bits of the block maximum, whether the converts
    apply (msq_quant_lowp.hip mx_setup_e4m3_lean) -- scales at either end of T's range, the range in which `scale + 1e-6` is no longer the
    scale, the floor(log2) bump just under powers of two, zero / Inf / NaN blocks must all come out as the reference's in-dtype arithmetic
    (oracle: msq_oracle_quantize_mx_lowp).  One block maximum per binade of T (plus its bump neighbours), values spread below it; keys
    layout (blocks along tokens, two channels per lane) and values layout (blocks along the contiguous axis, 8 values per lane).
    Round 6: the same for the e2m1 convert path (MX-FP4), whose zero tie sits a factor 16 under the maximum, not 2^-6 of it."""
    DEEP = 2.0 ** -6 if FMT == "fp8_e4m3" else 2.0 ** -3
    from msq import kvcache
    dn = "f16" if dtype == torch.float16 else "bf16"
    g = torch.Generator().manual_seed(77)
    exps = list(range(-26, 17)) if dtype == torch.float16 else list(range(-135, 128, 3)) + [-133, -126, -127, 126, 127]
    tops = []
    for e in exps:
        for mant in (1.0, 1.5, 2.0 - 2.0 ** -9, 2.0 - 2.0 ** -7, 2.0 - 2.0 ** -4):
            tops.append(mant * 2.0 ** e)
    tops = torch.tensor(tops, dtype=torch.float64).to(dtype)             # rounding to T: overflow -> inf, underflow -> 0 / subnormals: wanted
    nb = tops.numel()
    # values layout: [1, 1, nb, 128] = 4 blocks of 32 per row, each row under one maximum
    frac = torch.rand(nb, 128, generator=g, dtype=torch.float64) * 2 - 1
    frac[:, ::32] = 1.0
    frac[torch.rand(nb, 128, generator=g) < 0.2] *= DEEP                   # elements deep in the subnormal range / under the zero tie
    V_ = (frac * tops.double()[:, None]).to(dtype)
    V_[V_ != V_] = 0                                                       # inf * 0 of the overflowed maxima
    V_ = V_.reshape(1, 1, nb, 128)
    vq = kvcache.mx_quantize_values(V_.to(dev()), FMT, 32)
    vo = O.quantize_mx_lowp(V_.float().numpy(), dn, 8, FMT, 3, 32)
    assert _eq_bits(vq.float().cpu().numpy(), vo), dn
    # keys layout: [1, 1, 32 * nbk, 128]: channel c of token block b sits under maximum tops[(b * 128 + c) % nb]
    nbk = (nb + 127) // 128 + 1
    idx = (torch.arange(nbk)[:, None] * 128 + torch.arange(128)[None, :]) % nb
    top_k = tops.double()[idx]                                              # [nbk, 128]
    fk = torch.rand(nbk, 32, 128, generator=g, dtype=torch.float64) * 2 - 1
    fk[:, 5, :] = -1.0
    fk[torch.rand(nbk, 32, 128, generator=g) < 0.2] *= DEEP
    K_ = (fk * top_k[:, None, :]).to(dtype)
    K_[K_ != K_] = 0
    K_ = K_.reshape(1, 1, nbk * 32, 128)
    ko = O.quantize_mx_lowp(K_.float().numpy(), dn, 8, FMT, 2, 32)
    from msq._lib import lib
    try:
        for pair4 in (1, 0):                    # k_mx_lowp_pair4 (a block row over four waves, the default) and k_mx_lowp_pair
            assert lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", pair4) == 0
            kq = kvcache.mx_quantize_keys(K_.to(dev()), FMT, 32)
            assert _eq_bits(kq.float().cpu().numpy(), ko), (dn, pair4)
            # head dims that leave dead lanes (80: 40 channel pairs) and need two 64-pair chunks (160), other element formats
            for hd, fmt in ((80, "fp8_e4m3"), (160, "fp8_e4m3"), (128, "fp4_e2m1"), (80, "fp4_e2m1"), (160, "fp4_e2m1"), (160, "fp6_e3m2"), (80, "int8")):
                Kh = (torch.randn(2, 3, 64, hd, generator=g) * 3).to(dtype)
                Kh[0, 0, :32, 1] *= 2.0 ** -12
                got = kvcache.mx_quantize_keys(Kh.to(dev()), fmt, 32)
                want = O.quantize_mx_lowp(Kh.float().numpy(), dn, 8, fmt, 2, 32)
                assert _eq_bits(got.float().cpu().numpy(), want), (dn, pair4, hd, fmt)
    finally:
        lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", 1)
    # NaN and Inf members, and an all-zero block
    Z = V_.clone().reshape(nb, 128)
    Z[0, :32] = 0; Z[1, 3] = float("inf"); Z[2, 40] = float("nan"); Z[3, 70] = -float("inf")
    Z = Z.reshape(1, 1, nb, 128)
    zq = kvcache.mx_quantize_values(Z.to(dev()), FMT, 32)
    zo = O.quantize_mx_lowp(Z.float().numpy(), dn, 8, FMT, 3, 32)
    assert _eq_bits(zq.float().cpu().numpy(), zo), dn
    zk = kvcache.mx_quantize_keys(Z.reshape(1, 1, nb * 4, 32).repeat(1, 1, 1, 4)[:, :, :(nb * 4 // 32) * 32].contiguous().to(dev()), FMT, 32)
    zko = O.quantize_mx_lowp(Z.reshape(1, 1, nb * 4, 32).repeat(1, 1, 1, 4)[:, :, :(nb * 4 // 32) * 32].contiguous().float().numpy(), dn, 8, FMT, 2, 32)
    assert _eq_bits(zk.float().cpu().numpy(), zko), dn


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kv_mx_fp4_every_tie_and_the_excepted_magnitude(msq, O, dtype):
    """MX-FP4 through the e2m1 converts (round 6): every grid point and every tie of the format at every scale, two ulps of T either side -- the
    excepted magnitude pred_T(2^(se - 2)) among them, which the kernel lifts over the tie on its way into the convert --, both signs, against the
    oracle's op-by-op in-dtype arithmetic; values layout (k_mx_lowp_vec), keys layout through both strided kernels."""
    from msq import kvcache
    from msq._lib import lib
    dn = "f16" if dtype == torch.float16 else "bf16"
    g = torch.Generator().manual_seed(5)
    exps = list(range(-22, 13)) if dtype == torch.float16 else list(range(-120, 121, 7))
    rows = []
    for e in exps:
        k = torch.randint(0, 49, (4, 128), generator=g).double()             # k / 8: 0 ... 6 in steps of 1/8 (grid points: 0, .5, 1, 1.5, 2, 3, 4, 6; ties between)
        k[:, ::32] = 48                                                       # the block maximum 6 x 2^e: se = e
        k[:, 1::32] = 2                                                       # 0.25 x 2^e: the first tie; bumped below: the excepted magnitude
        v = (k / 8.0) * 2.0 ** e * (torch.randint(0, 2, (4, 128), generator=g).double() * 2 - 1)
        rows.append(v)
    X = torch.cat(rows).to(dtype)
    bits = X.view(torch.int16)
    bump = torch.randint(-2, 3, bits.shape, generator=g).to(torch.int16)
    bump[:, ::32] = 0                                                          # (the maxima stay: the scale is what the row is about)
    bump[:, 1::32] = -1                                                        # pred_T(first tie) in every block
    keep = (bits & 0x7FFF) < 4
    X = torch.where(keep, bits, bits + bump).view(dtype)
    nrow = X.shape[0]
    V_ = X.reshape(1, 1, nrow, 128)
    vq = kvcache.mx_quantize_values(V_.to(dev()), "fp4_e2m1", 32)
    vo = O.quantize_mx_lowp(V_.float().numpy(), dn, 8, "fp4_e2m1", 3, 32)
    assert _eq_bits(vq.float().cpu().numpy(), vo), dn
    # keys: the 32 values of a block along tokens -- transpose the 32-wide blocks of X into [tokens, channels]
    nbk = (nrow * 4) // 128
    K_ = X.reshape(nrow * 4, 32)[:nbk * 128].reshape(nbk, 128, 32).permute(0, 2, 1).contiguous().reshape(1, 1, nbk * 32, 128)
    ko = O.quantize_mx_lowp(K_.float().numpy(), dn, 8, "fp4_e2m1", 2, 32)
    try:
        for pair4 in (1, 0):
            assert lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", pair4) == 0
            kq = kvcache.mx_quantize_keys(K_.to(dev()), "fp4_e2m1", 32)
            assert _eq_bits(kq.float().cpu().numpy(), ko), (dn, pair4)
    finally:
        lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", 1)


# ----------------------------------------------------------------------------------------------------------------------
# activation producers in front of the MX Linear (RMSNorm, silu x up) and their fused MX-FP8 pack
# ----------------------------------------------------------------------------------------------------------------------
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bf_max_norm(bfloat):
    m = bfloat - 7
    return 2.0 ** 127 * (2 ** (m - 1) - 1) / 2 ** (m - 2)


def _decode_pack(codes, scales):
    """(codes [M, K] e4m3 bytes, scales [M, K / 32] bytes) -> float64 values"""
    c = codes.cpu().numpy(); s = scales.cpu().numpy().astype(np.float64)
    return _e4m3_decode(c) * np.repeat(np.exp2(s - 127.0), 32, axis=1)


def test_activation_producers_match_the_reference(msq, O):
    """mx.RMSNorm (layernorm.py:177), mx.silu (activations.py:76), mx.simd_mul (simd_ops.py:445) as single launches against the reference's
    CPU outputs (tests/golden/vec_ops2.npz): RMSNorm and simd_mul bit-exact (the row sum follows ATen's order) for both rounding specs and
    hidden sizes 128 / 200 (generic kernel) / 1024 / 4096 (register kernel), with and without bias; silu: the device expf replaces Sleef's,
    at most 2 elements per tensor one unit of the rounded format away (none seen)."""
    z = np.load(os.path.join(GOLD, "vec_ops2.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    for sn, sp in (("fp8_bf16", {"bfloat": 16}), ("bf12_even", {"bfloat": 12, "round": "even"})):
        specs = msq.specs.finalize_mx_specs(dict({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8,
                                                  "block_size": 32, "custom_cuda": True}, **sp))
        for H in (128, 200, 1024, 4096):
            k = f"rms|{sn}|{H}|"
            rn = msq.RMSNorm(H, eps=float(z[k + "eps"]), mx_specs=specs).to(dev())
            with torch.no_grad():
                rn.weight.copy_(t(z[k + "w"])); rn.bias.copy_(t(z[k + "b"]))
            assert _eq_bits(rn(t(z[k + "x"])).cpu().numpy(), z[k + "y"]), (sn, H)
            y0 = msq.vector_ops.rms_norm(t(z[k + "x"]), rn.weight, None, rn.eps, specs)
            assert _eq_bits(y0.cpu().numpy(), z[k + "y_nobias"]), (sn, H)
        k = f"act|{sn}|"
        s = msq.silu(t(z[k + "gate"]), mx_specs=specs).cpu().numpy()
        bad = s != z[k + "silu"]
        assert bad.sum() <= 2, (sn, int(bad.sum()))
        assert (np.abs(s - z[k + "silu"])[bad] <= np.abs(z[k + "silu"][bad]) * 2.0 ** -(7 if sn == "fp8_bf16" else 3)).all()
        assert _eq_bits(msq.simd_mul(t(z[k + "gate"]), t(z[k + "up"]), mx_specs=specs).cpu().numpy(), z[k + "mul"]), sn
        m = msq.vector_ops.silu_mul(t(z[k + "gate"]), t(z[k + "up"]), specs).cpu().numpy()
        bad = m != z[k + "silu_mul"]
        assert bad.sum() <= 2, (sn, int(bad.sum()))
        # silu x up on the reference's own silu values: the multiply is exact arithmetic, so this one is bit for bit
        assert _eq_bits(msq.simd_mul(t(z[k + "silu"]), t(z[k + "up"]), mx_specs=specs).cpu().numpy(), z[k + "silu_mul"]), sn
    assert torch.equal(msq.simd_mul(torch.ones(3, device=dev()), 2.0), torch.full((3,), 2.0, device=dev()))     # no specs: torch
    with pytest.raises(msq._lib.MsqError):
        msq.silu(torch.ones(4), mx_specs=specs)


def test_fused_producers_pack_what_the_unfused_chain_packs(msq, O):
    """msq_vec_rmsnorm_mx_pack_a8 / msq_vec_silu_mul_mx_pack_a8: the packed MX-FP8 operand (codes + scale bytes) equals, byte for byte,
    qlinear.mx_pack_act of the producer's own float32 output, for the register kernel (H = 1024, 4096, 8192), the generic one (H = 384, 640),
    strided gate / up halves of one [M, 2 I] tensor, and rows that hold Inf / NaN / zeros / subnormals; the float32 output, when asked for,
    is the producer's; the decoded operand equals the oracle's quantize_mx (OCP rule = the reference's native kernel, as every packer of the
    library) of the reference's RMSNorm output; against the reference's Python-path values (`+ 1e-6`, D2 of DESIGN.md) it differs where that
    rule moves ties (one grid step, ~1 / 16 of bfloat16-rounded data; tests/test_oracle_golden.py pins the Python-path values on the oracle)."""
    from msq import qlinear, vector_ops as V
    z = np.load(os.path.join(GOLD, "vec_ops2.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32,
                                         "custom_cuda": True, "bfloat": 16})
    g = torch.Generator(device=dev()).manual_seed(9)
    for H in (384, 640, 1024, 4096, 8192):
        for rows in (1, 7, 300):
            x = torch.randn(rows, H, device=dev(), generator=g) * 3
            x[0, :5] = torch.tensor([0.0, -0.0, 1e-40, 3e38, -7.0], device=dev())
            if rows > 2:
                x[1, 3] = float("inf"); x[2, 9] = float("nan")
            w = torch.randn(H, device=dev(), generator=g) * 0.3 + 1
            b = torch.randn(H, device=dev(), generator=g) * 0.1
            for bias in (b, None):
                y = V.rms_norm(x, w, bias, 1e-6, specs)
                c0, s0 = qlinear.mx_pack_act(y)
                (c1, s1), y1 = V.rms_norm_mx_pack(x, w, bias, 1e-6, specs, return_out=True)
                c2, s2 = V.rms_norm_mx_pack(x, w, bias, 1e-6, specs)
                assert torch.equal(c0, c1) and torch.equal(s0, s1) and torch.equal(c0, c2) and torch.equal(s0, s2), (H, rows, bias is None)
                assert _eq_bits(y.cpu().numpy(), y1.cpu().numpy()), (H, rows)
    # many rows: two and four rows per block (rows >= 2048 / 8192), against the oracle as well
    for rows in (2051, 8195):
        x = torch.randn(rows, 1024, device=dev(), generator=g) * 2
        w = torch.randn(1024, device=dev(), generator=g) * 0.3 + 1
        y = V.rms_norm(x, w, None, 1e-6, specs)
        assert _eq_bits(y.cpu().numpy(), O.vec_rmsnorm(x.cpu().numpy(), w.cpu().numpy(), np.zeros(1024, np.float32), 1e-6, 9, 8, _bf_max_norm(16), "nearest")), rows
        c0, s0 = qlinear.mx_pack_act(y)
        c1, s1 = V.rms_norm_mx_pack(x, w, None, 1e-6, specs)
        assert torch.equal(c0, c1) and torch.equal(s0, s1), rows
    for (M, I) in ((5, 128), (64, 1408), (33, 11008)):
        gu = torch.randn(M, 2 * I, device=dev(), generator=g) * 2.5
        gu[0, :4] = torch.tensor([0.0, -0.0, 90.0, -104.0], device=dev())
        gate, up = gu[:, :I], gu[:, I:]
        ref = V.silu_mul(gate.contiguous(), up.contiguous(), specs)
        assert _eq_bits(V.silu_mul(gate, up, specs).cpu().numpy(), ref.cpu().numpy()), (M, I)                 # strided halves, in place
        assert _eq_bits(ref.cpu().numpy(), V.simd_mul(V.silu(gate, mx_specs=specs), up, mx_specs=specs).cpu().numpy()), (M, I)
        c0, s0 = qlinear.mx_pack_act(ref)
        (c1, s1), o1 = V.silu_mul(gate, up, specs, pack=True, return_out=True)
        c2, s2 = V.silu_mul(gate, up, specs, pack=True)
        assert torch.equal(c0, c1) and torch.equal(s0, s1) and torch.equal(c0, c2) and torch.equal(s0, s2), (M, I)
        assert _eq_bits(o1.cpu().numpy(), ref.cpu().numpy())
    # against the reference-made fixture: RMSNorm output -> MX-FP8
    for H in (1024, 4096):
        k = f"rms|fp8_bf16|{H}|"
        c, s = V.rms_norm_mx_pack(t(z[k + "x"]), t(z[k + "w"]), t(z[k + "b"]), float(z[k + "eps"]), specs)
        dec = _decode_pack(c, s)
        want = O.quantize_mx(z[k + "y"], 8, "fp8_e4m3", axis=-1, block_size=32).astype(np.float64)
        assert (dec == want).all(), H
        # the reference's Python path divides by 2^e + 1e-6: every TIE of the scaled element moves down for scales below 2^5, and a
        # bfloat16-rounded activation (8 significant bits) is a tie of the 4-bit e4m3 grid once in sixteen -- D2 is not rare on this data
        rate = (dec != z[k + "y_mx"].astype(np.float64)).mean()
        assert 0.03 < rate < 0.09, (H, rate)
        assert (np.abs(dec - z[k + "y_mx"]) <= np.abs(want) * 2.0 ** -3 + 1e-30).all(), H          # ... by one step of the grid
    # the packed operand drives the GEMM: same result as handing the float32 activation to qlinear_mx_w4a8
    H, N = 4096, 512
    W = torch.randn(N, H, device=dev(), generator=g) * 0.02
    P = qlinear.mx_pack_weight(W)
    x = torch.randn(40, H, device=dev(), generator=g)
    w = torch.ones(H, device=dev())
    y_a = qlinear.qlinear_mx_w4a8(V.rms_norm(x, w, None, 1e-6, specs), P, None, torch.float32)
    y_b = qlinear.qlinear_mx_w4a8(V.rms_norm_mx_pack(x, w, None, 1e-6, specs), P, None, torch.float32)
    assert torch.equal(y_a, y_b)
    with pytest.raises(msq._lib.MsqError):
        V.rms_norm_mx_pack(torch.randn(4, 200, device=dev()), torch.ones(200, device=dev()), None, 1e-6, specs)     # H % 128


def test_gated_mlp_block_from_mx_modules_matches_the_reference(msq):
    """A Llama-style gated MLP written against the reference's `mx` module set -- RMSNorm -> MXLinear gate / up -> simd_mul(silu(gate), up) ->
    MXLinear down -> simd_add(residual, .), w fp4_e2m1 / a fp8_e4m3 / block 32 / bfloat 16 -- with the reference's parameters on its input
    (tests/golden/vec_ops2.npz `gmlp|*`): every stage against the reference's CPU intermediate.  RMSNorm, simd_mul, simd_add are exact;
    silu is exact up to the device expf (<= 2 elements); the MXLinears re-round an fp32 GEMM with another summation order to bfloat16
    (one bf16 ulp on <= 1 % there, as the ResidualMLP test of round 2)."""
    z = np.load(os.path.join(GOLD, "vec_ops2.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32,
                                      "bfloat": 16, "custom_cuda": True})

    class GatedMLP(torch.nn.Module):
        def __init__(self, hidden, inter, mx_specs):
            super().__init__()
            self.mx_specs = mx_specs
            self.norm = msq.RMSNorm(hidden, eps=1e-6, mx_specs=mx_specs)
            self.gate_proj = msq.MXLinear(hidden, inter, bias=False, mx_specs=mx_specs)
            self.up_proj = msq.MXLinear(hidden, inter, bias=False, mx_specs=mx_specs)
            self.down_proj = msq.MXLinear(inter, hidden, bias=False, mx_specs=mx_specs)

        def forward(self, x):
            x, residual = msq.simd_split(x)
            h = self.norm(x)
            a = msq.simd_mul(msq.silu(self.gate_proj(h), mx_specs=self.mx_specs), self.up_proj(h), mx_specs=self.mx_specs)
            return msq.simd_add(residual, self.down_proj(a), mx_specs=self.mx_specs)

    blk = GatedMLP(128, 384, sp).to(dev())
    blk.load_state_dict({k[len("gmlp|param|"):]: t(z[k]) for k in z.files if k.startswith("gmlp|param|")})
    x = t(z["gmlp|x"])

    def close(a, ref, what, frac=0.01):
        e = np.abs(a.cpu().numpy() - ref)
        assert (e <= np.abs(ref) * 2.0 ** -7 + 1e-6).all() and (e > 0).mean() <= frac, (what, float(e.max()), float((e > 0).mean()))
    with torch.no_grad():
        assert _eq_bits(blk.norm(x).cpu().numpy(), z["gmlp|norm"])
        close(blk.gate_proj(t(z["gmlp|norm"])), z["gmlp|gate"], "gate")
        close(blk.up_proj(t(z["gmlp|norm"])), z["gmlp|up"], "up")
        a = msq.vector_ops.silu_mul(t(z["gmlp|gate"]), t(z["gmlp|up"]), sp)
        assert (a.cpu().numpy() != z["gmlp|act"]).sum() <= 2
        close(blk.down_proj(t(z["gmlp|act"])), z["gmlp|down"], "down")
        assert _eq_bits(msq.simd_add(x, t(z["gmlp|down"]), mx_specs=sp).cpu().numpy(), z["gmlp|y"])
        y = blk(x).cpu().numpy()
    e = np.abs(y - z["gmlp|y"])
    assert (e <= np.abs(z["gmlp|y"]) * 2.0 ** -6 + 1e-6).all(), float(e.max())
    assert (e > 0).mean() <= 0.05, float((e > 0).mean())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_producers_read_16_bit_activations_as_they_are(msq, dtype):
    """msq_vec_rmsnorm_mx_pack_a8_x16 / msq_vec_silu_mul_mx_pack_a8_x16: a float16 / bfloat16 activation goes in without a cast pass and gives
    the bytes (and, when asked for, the float32 output) of the float32 entry on x.float() -- every 16-bit value is a float32 value; register
    and generic RMSNorm kernels, contiguous and strided gate / up halves, a strided view that is not 16-byte aligned (falls back)."""
    from msq import vector_ops as V
    specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32,
                                         "custom_cuda": True, "bfloat": 16})
    g = torch.Generator(device=dev()).manual_seed(31)
    for H in (384, 4096, 8192):
        x = (torch.randn(37, H, device=dev(), generator=g) * 4).to(dtype)
        x[0, :3] = torch.tensor([0.0, -0.0, 6e-8], device=dev()).to(dtype)
        w = torch.randn(H, device=dev(), generator=g) * 0.3 + 1
        (c0, s0), y0 = V.rms_norm_mx_pack(x.float(), w, None, 1e-5, specs, return_out=True)
        (c1, s1), y1 = V.rms_norm_mx_pack(x, w, None, 1e-5, specs, return_out=True)
        assert torch.equal(c0, c1) and torch.equal(s0, s1) and _eq_bits(y0.cpu().numpy(), y1.cpu().numpy()), H
    for (M, I) in ((9, 256), (40, 11008)):
        gu = (torch.randn(M, 2 * I, device=dev(), generator=g) * 3).to(dtype)
        for gate, up in ((gu[:, :I], gu[:, I:]), (gu[:, :I].contiguous(), gu[:, I:].contiguous())):
            (c0, s0), y0 = V.silu_mul(gate.float(), up.float(), specs, pack=True, return_out=True)
            (c1, s1), y1 = V.silu_mul(gate, up, specs, pack=True, return_out=True)
            assert torch.equal(c0, c1) and torch.equal(s0, s1) and _eq_bits(y0.cpu().numpy(), y1.cpu().numpy()), (M, I)
    odd = (torch.randn(8, 2 * 256 + 4, device=dev(), generator=g)).to(dtype)
    ga, ub = odd[:, 2:258], odd[:, 258:514]                      # 4-byte aligned views: the wrapper takes the float32 entry
    assert _eq_bits(V.silu_mul(ga, ub, specs).cpu().numpy(), V.silu_mul(ga.float(), ub.float(), specs).cpu().numpy())


def test_vector_rounding_under_truncation_follows_the_python_path(msq, O):
    """quantize_elemwise_op with round="floor" (elemwise_ops.py:139-140, :47-78): the private exponent is floor(torch.log2(|x|)) in float32,
    one binade high for the K largest floats below a power of two, and truncation then drops one more mantissa bit.  (a) the rounding of the
    vector ops (msq_vec_round, bfloat16 / 12 / 10, all three modes) against the oracle's Python-path core on the 55 k probes of
    tests/golden/log2_f32.npz (the 128 largest floats below every power of two, both signs); (b) RMSNorm and simd_add against fixtures made by
    the reference itself (tests/golden/vec_rmsnorm_modes.npz, make_golden_vec_modes.py: bfloat12 / 16, floor / nearest, inputs scaled 2^-30 ..
    2^30, eps 1e-6 / 1e-12 -- `b - tiny` under floor lands on exactly those floats)."""
    from msq import vector_ops as V
    L = msq._lib.lib()
    z = np.load(os.path.join(GOLD, "log2_f32.npz"))
    xs = z["bits"].astype(np.uint32).view(np.float32)
    # (not the 44 largest floats: there torch.log2 says 128, the Python path scales by 2^128 = inf and returns NaN in every mode; the
    # library returns +-inf on overflow, as the reference's native kernel does)
    xs = xs[np.isfinite(xs) & (xs > 0) & (xs < 3.4e38)]
    x = np.concatenate([xs, -xs]).astype(np.float32)
    xt = torch.from_numpy(x).to(dev())
    for bfloat in (16, 12, 10):
        m = bfloat - 7
        mn = 2.0 ** 127 * (2 ** (m - 1) - 1) / 2 ** (m - 2)
        for rm, rd in ((0, "nearest"), (1, "floor"), (2, "even")):
            out = torch.empty_like(xt)
            msq._lib.check(L.msq_vec_round(msq._lib.ptr(xt), msq._lib.ptr(out), xt.numel(), m, 8, mn, rm, 1, 0, msq._lib.current_stream(dev())), "msq_vec_round")
            want = O.quantize_elemwise_core(x, m, 8, mn, rd, False, True)
            assert _eq(out.cpu().numpy(), want), (bfloat, rd, int((out.cpu().numpy() != want).sum()))
    zz = np.load(os.path.join(GOLD, "vec_rmsnorm_modes.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    for k in sorted(set(f.rsplit("|", 1)[0] for f in zz.files)):
        bf, rd, H, sc, eps = k.split("|")
        specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32,
                                             "custom_cuda": True, "bfloat": int(bf), "round": rd})
        y = V.rms_norm(t(zz[k + "|x"]), t(zz[k + "|w"]), t(zz[k + "|b"]), float(eps), specs).cpu().numpy()
        assert _eq(y, zz[k + "|y"]), (k, int((y != zz[k + "|y"]).sum()))
