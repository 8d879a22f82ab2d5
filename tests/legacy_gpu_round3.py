"""Round-3 GPU tests: the staggered eight-wave GEMM schedule, shapes off the kernels' tile grid, half-precision packing that
matches the RTN harness, fused q / k / v and gate / up projections, HIP-graph capture of a packed decode step, GPTQ options."""
import os
import types

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


def _weights(N, K, seed=0):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, K, generator=g) * 0.02
    W[torch.rand(N, K, generator=g) < 0.01] *= 16
    return W


# ----------------------------------------------------------------------------------------------------------------------
# staggered schedule of k_qgemm3 (waves 4-7 half a K-step behind): K-step counts around the prologue / tail special cases
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [64, 128, 192, 320, 1088])
@pytest.mark.parametrize("fo", ["posit8_es1", "fp8_e4m3"])
def test_staggered_gemm_short_and_odd_k(msq, fo, K):
    """The eight-wave blocks need (M / 128) (N / 512) >= 256: M 2048 x N 8192.  The posit layout runs the staggered schedule
    (role barriers, four activation buffers), the fp8 layout the plain one: both must equal the dense product of the
    unpacked weight (fp32 accumulation: 2e-5 max|y|) for 1, 2, 3, 5 and 17 K-steps, with ragged M, and repeat bit for bit."""
    M, N = 2048 - 37, 8192 + 8192
    W = _weights(N, K, 1).to(dev())
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(2)).to(dev()).to(torch.bfloat16)
    P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    ref = X.float() @ msq.qlinear.unpack_weight(P).t()
    y0 = msq.qlinear.qlinear(X, P, None, torch.float32)
    assert (y0 - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
    for _ in range(20):
        assert torch.equal(msq.qlinear.qlinear(X, P, None, torch.float32), y0)


# ----------------------------------------------------------------------------------------------------------------------
# shapes off the tile grid: padded at pack time, sliced on the way out (utils/modelutils.py:8-15: every Linear is quantised)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K", [(1000, 520), (300, 96), (257, 4100)])
def test_offgrid_pack_matches_oracle(msq, O, N, K):
    W = _weights(N, K, 3)
    X = torch.randn(70, K, generator=torch.Generator().manual_seed(4))
    xb = X.to(dev()).to(torch.bfloat16)
    for fo, layout in (("fp8_e4m3", "unified"), ("posit8_es1", "unified"), ("fp8_e4m3", "planes")):
        o = O.outlier_fakequant(W.numpy(), 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"]
        P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
        assert (P.n, P.k) == (N, K) and P.N % 256 == 0 and P.K % 64 == 0
        Wq = msq.qlinear.unpack_weight(P).cpu().numpy()
        assert Wq.shape == (N, K) and (Wq == o).all(), (fo, layout)
        for rows in (70, 5):
            y = msq.qlinear.qlinear(xb[:rows], P, None, torch.float32).cpu().numpy()
            yr = O.linear(xb[:rows].float().cpu().numpy(), o)
            assert y.shape == (rows, N) and np.abs(y - yr).max() <= 2e-5 * np.abs(yr).max() + 1e-6
    # values packed as they are, with a bias and a caller-provided output buffer
    o = O.outlier_fakequant(W.numpy(), 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    Pv = msq.qlinear.pack_values(torch.from_numpy(o).to(dev()))
    b = torch.randn(N, generator=torch.Generator().manual_seed(5)).to(dev())
    out = torch.empty(70, N, dtype=torch.float32, device=dev())
    y = msq.qlinear.qlinear(xb, Pv, b, torch.float32, out=out)
    assert y.data_ptr() == out.data_ptr()
    yr = O.linear(xb.float().cpu().numpy(), o, b.cpu().numpy())
    assert np.abs(y.cpu().numpy() - yr).max() <= 2e-5 * np.abs(yr).max() + 1e-6
    # MX matrix path: plain MX-FP4 weights x MX-FP8 activations on the padded grid == the oracle's quantisers on the logical shape
    Wm = O.quantize_mx(W.numpy(), 8, "fp4_e2m1", axis=-1, block_size=32)
    Xm = O.quantize_mx(X.numpy(), 8, "fp8_e4m3", axis=-1, block_size=32)
    Pm = msq.qlinear.mx_pack_weight(W.to(dev()))
    assert (Pm.n, Pm.k) == (N, K) and Pm.K % 128 == 0
    ym = msq.qlinear.qlinear_mx_w4a8(X.to(dev()), Pm, None, torch.float32).cpu().numpy()
    yr = O.linear(Xm, Wm)
    assert ym.shape == (70, N) and np.abs(ym - yr).max() <= 1e-4 * np.abs(yr).max() + 1e-6


def test_offgrid_model_is_fully_packed(msq, tmp_path):
    """A model whose Linears fit no tile (hidden 200, intermediate 520): pack_layers leaves nothing dense, on both paths, the
    packed model reproduces the fake-quantised one, and the checkpoint of the padded planes round-trips."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from msq.harness.data_utils import _Enc
    from msq.harness.evalppl import pack_layers, perplexity, quantize_layers_nearest
    from msq import checkpoint as ckpt
    cfg = LlamaConfig(hidden_size=200, intermediate_size=520, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=512, max_position_embeddings=128)
    tokens = _Enc(torch.randint(0, 512, (1, 64 * 4), generator=torch.Generator().manual_seed(1)))
    for path in ("bf16", "mx"):
        torch.manual_seed(0)
        m = LlamaForCausalLM(cfg).eval().to(dev())
        quantize_layers_nearest(m.model.layers, dev(), dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3",
                                                            axes=[-1], block_size=32))
        Wq = m.model.layers[1].mlp.down_proj.weight.data.clone()
        ppl_dense = perplexity(m, tokens, dev(), 64)
        packed, dense = pack_layers(m.model.layers, path=path)
        assert (packed, dense) == (14, 0)
        assert not any(isinstance(l, torch.nn.Linear) for layer in m.model.layers for l in layer.modules())
        dp = m.model.layers[1].mlp.down_proj
        assert (dp.in_features, dp.out_features) == (520, 200)
        if path == "bf16":
            assert torch.equal(dp.dequantize(), Wq)
        ppl = perplexity(m, tokens, dev(), 64)
        assert abs(ppl - ppl_dense) / ppl_dense < (0.02 if path == "bf16" else 0.05), (path, ppl, ppl_dense)
        f = str(tmp_path / ("offgrid_%s.safetensors" % path))
        ckpt.save_packed(m, f)
        torch.manual_seed(0)
        m2 = LlamaForCausalLM(cfg).eval().to(dev())
        ckpt.load_packed(m2, f)
        x = torch.randint(0, 512, (1, 16), device=dev())
        assert torch.equal(m(x).logits, m2(x).logits)


# ----------------------------------------------------------------------------------------------------------------------
# half-precision checkpoints: the packed layer holds exactly what the RTN harness writes into the weight (advisor, round 2)
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_from_linear_on_half_weight_equals_rtn_values(msq, dtype):
    g = torch.Generator().manual_seed(7)
    lin = torch.nn.Linear(512, 768, bias=True)
    lin.weight.data = torch.randn(768, 512, generator=g) * 0.03      # explicit: get_llama / get_opt swap torch's initialisers out
    lin.weight.data[torch.rand(768, 512, generator=g) < 0.01] *= 12
    lin.bias.data = torch.randn(768, generator=g) * 0.1
    lin = lin.to(dev()).to(dtype)
    q = msq.quant.MXQuantizer()
    q.configure(8, 8, inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3", axes=[-1], block_size=32)
    rtn = q.quantize(lin.weight.data)                       # what llm/llama.py:238 stores (computed in `dtype`, op by op)
    ql = msq.qlinear.QuantLinear.from_linear(lin, q)
    assert torch.equal(ql.dequantize(), rtn.float())
    # ... and differs from the upcast route on a few weights (that is the point of threading compute_dtype through)
    ql32 = msq.qlinear.QuantLinear.from_linear(lin, q, compute_dtype="float32")
    assert ql32.dequantize().shape == rtn.shape
    x = torch.randn(9, 512, device=dev(), dtype=dtype)
    y = ql(x)
    assert y.dtype == dtype
    ref = torch.nn.functional.linear(x.to(torch.bfloat16).float(), rtn.float(), lin.bias.float())
    assert (y.float() - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item()


# ----------------------------------------------------------------------------------------------------------------------
# fused q / k / v and gate / up projections (make_quant(fuse=...)) + HIP-graph capture of a decode step
# ----------------------------------------------------------------------------------------------------------------------
def _tiny_llama(dtype=torch.float32, hidden=256, inter=512, layers=2):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=512, max_position_embeddings=128)
    torch.manual_seed(0)
    return LlamaForCausalLM(cfg).eval().to(dtype).to(dev())


@pytest.mark.parametrize("path", ["bf16", "mx"])
def test_fused_projections_same_values_fewer_launches(msq, tmp_path, path):
    from msq.harness import find_layers
    from msq.harness.evalppl import LLAMA_FUSE, pack_layers, quantize_layers_nearest
    from msq import checkpoint as ckpt
    qc = dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3", axes=[-1], block_size=32)
    a, b = _tiny_llama(), _tiny_llama()
    for m in (a, b):
        quantize_layers_nearest(m.model.layers, dev(), qc)
    assert pack_layers(a.model.layers, path=path) == (14, 0)
    assert pack_layers(b.model.layers, path=path, fuse=LLAMA_FUSE) == (14, 0)
    att = b.model.layers[0].self_attn
    assert all(isinstance(p, msq.qlinear.ProjectionSlice) for p in (att.q_proj, att.k_proj, att.v_proj))
    fused = att.q_proj.shared()
    assert att.k_proj.shared() is fused and fused.proj.out_features == 768 and fused.splits == [256, 256, 256]
    if path == "bf16":      # the values are those of the three separate weights
        Wf = fused.proj.dequantize()
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            assert torch.equal(Wf[256 * i:256 * (i + 1)], getattr(a.model.layers[0].self_attn, n).dequantize())
    # launches: one packed module per group
    n_packed = lambda m: sum(isinstance(l, (msq.qlinear.QuantLinear, msq.qlinear.MXLinearW4A8)) for l in m.modules())
    assert n_packed(a) == 14 and n_packed(b) == 8
    x = torch.randint(0, 512, (2, 24), device=dev())
    ya, yb = a(x).logits, b(x).logits
    assert (ya - yb).abs().max().item() <= 2e-2 * ya.abs().max().item()       # same weights; the fused GEMM sums in another order
    # the slices fall back to one GEMM per call when they are fed different tensors
    h = torch.randn(3, 256, device=dev())
    q1 = att.q_proj(h)
    k1 = att.k_proj(h.clone())
    assert q1.shape == (3, 256) and k1.shape == (3, 256)
    assert torch.equal(att.v_proj(h), fused.proj(h)[..., 512:768])
    # checkpoint round trip keeps the fusion
    f = str(tmp_path / "fused.safetensors")
    hdr = ckpt.save_packed(b, f)
    assert any("fused" in d for d in hdr["layers"].values())
    c = _tiny_llama()
    ckpt.load_packed(c, f)
    assert isinstance(c.model.layers[1].mlp.up_proj, msq.qlinear.ProjectionSlice)
    assert torch.equal(c(x).logits, yb)


@pytest.mark.parametrize("path", ["bf16", "mx"])
def test_packed_decode_step_is_graph_capturable(msq, path):
    """One decode step (M = 1) of a packed tiny Llama -- fused projections, QuantLinear / MXLinearW4A8 forwards through the
    public modules -- captured into a HIP graph and replayed: no host synchronisation, no allocation-dependent pointers;
    the replays equal the eager result bit for bit, also after the input buffer has been rewritten."""
    from msq.harness.evalppl import LLAMA_FUSE, pack_layers, quantize_layers_nearest
    m = _tiny_llama(torch.bfloat16)
    quantize_layers_nearest(m.model.layers, dev(), dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3",
                                                        axes=[-1], block_size=32))
    assert pack_layers(m.model.layers, path=path, fuse=LLAMA_FUSE) == (14, 0)
    layers = m.model.layers
    rot = m.model.rotary_emb
    h_in = torch.randn(1, 1, 256, device=dev(), dtype=torch.bfloat16)
    pos = torch.zeros(1, 1, dtype=torch.long, device=dev())

    def step(h):
        pe = rot(h, pos)
        for layer in layers:
            out = layer(h, position_embeddings=pe, attention_mask=None)
            h = out[0] if isinstance(out, tuple) else out
        return h

    with torch.no_grad():
        eager = step(h_in).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step(h_in)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = step(h_in)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, eager)
        h2 = torch.randn(1, 1, 256, device=dev(), dtype=torch.bfloat16)
        eager2 = step(h2).clone()
        h_in.copy_(h2)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, eager2)


# ----------------------------------------------------------------------------------------------------------------------
# GPTQ options (llm/gptq.py:81-87,119-127) and the block kernel's residency check
# ----------------------------------------------------------------------------------------------------------------------
def test_gptq_groupsize_and_static_groups_are_honoured(msq):
    from msq.harness.gptq import GPTQ
    torch.manual_seed(11)
    qc = dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3", axes=[0], block_size=16)

    def run(quantizer_cls=None, **kw):
        gen = torch.Generator().manual_seed(11)
        lin = torch.nn.Linear(256, 192, bias=False)
        lin.weight.data = torch.randn(192, 256, generator=gen) * 0.05
        lin = lin.to(dev())
        g = GPTQ(lin)
        X = torch.randn(4, 64, 256, generator=gen).to(dev())
        g.add_batch(X, lin(X))
        q = (quantizer_cls or msq.quant.MXQuantizer)()
        q.configure(8, 8, **qc)
        g.quantizer = q
        g.fasterquant(blocksize=128, percdamp=0.01, verbose=False, **kw)
        return lin.weight.data.clone(), g

    base, _ = run()
    # MXQuantizer.find_params is empty (utils/quant.py:429): groups change nothing, the block kernel stays in use
    for kw in (dict(groupsize=64), dict(groupsize=64, static_groups=True), dict(groupsize=32, static_groups=True, actorder=True)):
        w, _ = run(**kw)
        if "actorder" not in kw:
            assert torch.equal(w, base), kw
    # a quantiser whose find_params does something is re-fitted per group, column by column
    calls = []

    class Counting(msq.quant.MXQuantizer):
        def find_params(self, x, weight=False):
            calls.append(tuple(x.shape))

    w, _ = run(Counting, groupsize=64)
    assert calls.count((192, 64)) == 4 and torch.equal(w, base)       # one re-fit per 64-column group, same values
    calls.clear()
    w, _ = run(Counting, groupsize=128, static_groups=True)
    assert calls.count((192, 128)) == 2 and torch.equal(w, base)
    with pytest.raises(ValueError):
        run(static_groups=True)
    with pytest.raises(ValueError):
        run(groupsize=0)


def test_gptq_block_kernel_checks_residency(msq):
    """msq_gptq_block sizes its grid against the device (CUs x resident blocks at its LDS size) and refuses what cannot be
    co-resident; on a whole MI355X every legal grid (<= 200 workgroups) fits, and the status word carries no time-out bit."""
    from msq._lib import check, current_stream, lib, ptr
    from msq.formats import format_id
    L = lib()
    O_, cols = 51200, 128
    torch.manual_seed(2)
    Wt = torch.randn(cols, O_, device=dev()) * 0.02
    U = torch.eye(cols, device=dev()) + torch.triu(torch.randn(cols, cols, device=dev()) * 0.01, 1)
    Qt, Et = torch.empty_like(Wt), torch.empty_like(Wt)
    loss = torch.zeros((), dtype=torch.float64, device=dev())
    pruned = torch.zeros((), dtype=torch.int64, device=dev())
    status = torch.zeros(1, dtype=torch.int32, device=dev())
    wsb = L.msq_gptq_block_workspace_bytes(O_, cols)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
    check(L.msq_gptq_block(ptr(Wt), ptr(U), cols, ptr(Qt), ptr(Et), ptr(loss), ptr(pruned), ptr(status), ptr(ws), wsb, O_, cols, 16,
                           format_id("fp4_e2m1"), format_id("fp8_e4m3"), 8, 8, 2.0, 0, 0, current_stream(dev())), "msq_gptq_block")
    torch.cuda.synchronize()
    assert int(status.item()) & 4 == 0 and torch.isfinite(Qt).all()
    wsb2 = L.msq_gptq_block_workspace_bytes(51200 + 256, cols)          # 201 workgroups: refused before anything is launched
    ws2 = torch.empty(wsb2, dtype=torch.uint8, device=dev())
    rc = L.msq_gptq_block(ptr(Wt), ptr(U), cols, ptr(Qt), ptr(Et), ptr(loss), ptr(pruned), ptr(status), ptr(ws2), wsb2, 51200 + 256, cols, 16,
                          format_id("fp4_e2m1"), format_id("fp8_e4m3"), 8, 8, 2.0, 0, 0, current_stream(dev()))
    assert rc == -2


# ----------------------------------------------------------------------------------------------------------------------
# fp16 output straight from the kernels' epilogues (y_dtype 1): an fp16 model needs no cast pass over the result
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M", [1, 20, 48, 130, 300, 2048])
def test_fp16_output_equals_rounded_fp32_output(msq, M):
    """Every launch shape (decode kernels with and without the reduce kernel, split-K, 64-row tiles, k-groups, eight-wave blocks;
    bf16-activation and MX paths): the fp16 result is the fp32 result of the same launch rounded once to half."""
    N, K = (8192, 1024) if M == 2048 else (4096, 1024)
    W = _weights(N, K, 6).to(dev())
    b = torch.randn(N, generator=torch.Generator().manual_seed(8)).to(dev())
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(7)).to(dev())
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        y32 = msq.qlinear.qlinear(X, P, b, torch.float32)
        y16 = msq.qlinear.qlinear(X, P, b, torch.float16)
        assert y16.dtype == torch.float16 and torch.equal(y16, y32.to(torch.float16)), (fo, M)
        assert torch.equal(msq.qlinear.qlinear(X, P, b, torch.bfloat16), y32.to(torch.bfloat16))
    for Pm in (msq.qlinear.mx_pack_weight(W), msq.qlinear.mx_pack_weight(W, w_fmt="e3m2"),
               msq.qlinear.mx_pack_values(msq.quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])):
        y32 = msq.qlinear.qlinear_mx_w4a8(X, Pm, b, torch.float32)
        assert torch.equal(msq.qlinear.qlinear_mx_w4a8(X, Pm, b, torch.float16), y32.to(torch.float16)), (Pm.w_fmt, M)
    # the modules hand an fp16 model its dtype without a cast
    lin = torch.nn.Linear(K, N, bias=True)
    lin.weight.data, lin.bias.data = W.cpu(), b.cpu()
    q = msq.quant.MXQuantizer(); q.configure(8, 8, inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3", axes=[-1], block_size=32)
    ql = msq.qlinear.QuantLinear.from_linear(lin.to(dev()), q)
    assert ql(X.half()).dtype == torch.float16 and ql(X.bfloat16()).dtype == torch.bfloat16


# ----------------------------------------------------------------------------------------------------------------------
# bench.py --workload llama7b_e2e on a 4-layer slice of the Llama-2-7B-shaped model
# ----------------------------------------------------------------------------------------------------------------------
def test_bench_llama7b_e2e_slice(msq):
    """Whole-model evidence (llm/llama.py:176-284, llm/opt.py:332-376 at Llama-2-7B's true layer shapes, random weights): every key
    is there, nothing stays dense, the packed model is 9.25 bits per weight, and the perplexity of the packed model is within
    0.9 % of the fake-quantised dense one (= 0.05 at PPL 5.5, the north-star bound)."""
    import json, subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "llama7b_e2e", "--layers", "4", "--seqlen", "1024",
                          "--decode-tokens", "10", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["kept_dense"] == 0 and d["packed_linears"] == 28 and d["packed_modules"] == 16
    assert abs(d["packed_bits_per_weight"] - 9.25) < 1e-6 and d["packed_GB"] < 0.6 * d["dense_16bit_GB"]
    assert d["rtn_quantise_s"] > 0 and d["pack_s"] > 0
    for k in ("packed_fused", "fakequant_dense_fp16"):
        r = d["speed"][k]
        assert r["prefill_tokens_per_s"] > 0 and r["decode_ms_per_token_eager"] > 0 and r["decode_linears_ms_per_token_hip_graph"] > 0
    assert d["decode_linears_speedup_hip_graph"] > 1.0          # the packed weight stream is 9.25 bits against 16
    p = d["ppl_proxy_7b"]
    assert p["relative_delta"] <= 0.009, p
    assert p["max_logit_abs_err"] <= 0.05 * p["max_abs_logit"], p


# ----------------------------------------------------------------------------------------------------------------------
# the 70B row-parallel bench line on ONE GPU with a real (world-size-1) RCCL group: the evidence keys of the multi-GPU run
# ----------------------------------------------------------------------------------------------------------------------
def test_bench_rowparallel_evidence_on_single_rank_rccl_group(msq):
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "llama70b_rowparallel", "--single-rank-collectives",
                          "--steps", "10", "--warmup", "2", "--chunks", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 1
    seen = d["config"]["ranks_seen"]
    assert len(seen) == 1 and seen[0]["rank"] == 0 and seen[0]["world_size"] == 1 and seen[0]["backend"] == "nccl" and seen[0]["rccl"]
    r = d["rowparallel"]
    assert r["chunks"] == 2 and r["comm"] == "rs_ag" and r["gemm_ms"] > 0 and r["comm_ms"] > 0 and r["step_ms"] >= r["gemm_ms"] * 0.98
    assert abs(r["exposed_comm_ms"] - (r["step_ms"] - r["gemm_ms"])) < 1e-6 or r["exposed_comm_ms"] == 0.0


# ----------------------------------------------------------------------------------------------------------------------
# _quantize_mx on half tensors: one launch, computed in the tensor dtype like the reference (judge round 2, item 3b)
# ----------------------------------------------------------------------------------------------------------------------
def _bits(a, dn):
    t = torch.from_numpy(a.view(np.int16).copy())
    return t.view(torch.float16 if dn == "f16" else torch.bfloat16)


def test_quantize_mx_half_tensors_match_reference_goldens(msq):
    import json
    z = np.load(os.path.join(G, "quantize_mx_lowp.npz"))
    meta = json.load(open(os.path.join(G, "quantize_mx_lowp_meta.json")))
    n = 0
    for key, m in sorted(meta.items()):
        dn, tname, cname = key.split("|")
        sb, fmt, ax, bs, rnd, flush = m["cfg"]
        A = _bits(z[f"in|{dn}|{tname}"], dn).to(dev())
        y = msq.mx_ops._quantize_mx(A, sb, fmt, "max", ax, bs, rnd, flush)
        ref = _bits(z[f"out|{key}"], dn).to(dev())
        assert y.dtype == A.dtype
        same = (y.view(torch.int16) == ref.view(torch.int16)) | (torch.isnan(y) & torch.isnan(ref))
        assert bool(same.all()), (key, int((~same).sum()))
        n += 1
    assert n == 224


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_kv_cache_mx_variant_is_one_launch_in_the_cache_dtype(msq, O, dtype):
    """BASELINE config 4, MX variant of the cache: keys in blocks of 32 tokens of one channel, values in blocks of 32 channels of
    one token, straight on the fp16 / bf16 cache tensor; equal to the oracle's half-precision restatement on a [1, 4, 96, 128]
    cache and on a ragged one (S = 70), and the compress_insert_function hook writes the same values in place."""
    from msq import kvcache
    g = torch.Generator().manual_seed(5)
    for S in (96, 70):
        K_ = (torch.randn(1, 4, S, 128, generator=g) * 1.5).to(dtype)
        V_ = (torch.randn(1, 4, S, 128, generator=g) * 0.7).to(dtype)
        dn = "f16" if dtype == torch.float16 else "bf16"
        kq = kvcache.mx_quantize_keys(K_.to(dev()), "fp8_e4m3", 32)
        vq = kvcache.mx_quantize_values(V_.to(dev()), "fp8_e4m3", 32)
        ko = O.quantize_mx_lowp(K_.float().numpy(), dn, 8, "fp8_e4m3", 2, 32)
        vo = O.quantize_mx_lowp(V_.float().numpy(), dn, 8, "fp8_e4m3", 3, 32)
        assert kq.dtype == dtype and (kq.float().cpu().numpy() == ko).all() and (vq.float().cpu().numpy() == vo).all()


def test_simd_add_broadcasts_both_ways(msq):
    """mx.simd_add (simd_ops.py:80-106: "Shape broadcasting is fully supported"): an in1 of [1, H] plus an in2 of [B, S, H], the result
    in the broadcast shape and torch's promoted dtype (advisor, round 2)."""
    sp = msq.specs.finalize_mx_specs({"bfloat": 16, "custom_cuda": True})
    a = torch.randn(1, 64, device=dev())
    b = torch.randn(2, 5, 64, device=dev())
    y = msq.vector_ops.simd_add(a, b, sp)
    y2 = msq.vector_ops.simd_add(b, a, sp)
    assert y.shape == (2, 5, 64) and torch.equal(y, y2)
    ref = msq.vector_ops.simd_add(a.expand(2, 5, 64).contiguous(), b, sp)
    assert torch.equal(y, ref)
    assert msq.vector_ops.simd_add(a.half(), b, sp).dtype == torch.float32          # promotion, as `in1 + in2`
    assert msq.vector_ops.simd_add(a.half(), 1.5, sp).dtype == torch.float16


@pytest.mark.parametrize("M", [1, 9, 16, 32, 64, 65])
def test_decode_kernels_take_fp16_activations(msq, M):
    """An fp16 model at decode sizes: qlinear on the fp16 tensor (converted inside the weight-streaming kernels, msq_qlinear_f16x) ==
    qlinear on x.to(bfloat16), bit for bit, for every packed layout; M = 65 (and M = 64 on a wide layer) falls back to the cast."""
    torch.manual_seed(50 + M)
    for N, K in ((4096, 4096), (16384, 1024)):
        W = _weights(N, K, 12).to(dev())
        x = (torch.randn(M, K, device=dev()) * 2).half()
        x[0, 0] = 65504.0; x[M - 1, 1] = 6e-8          # the largest fp16 value and an fp16 subnormal
        b = torch.randn(N, device=dev())
        for fo, layout in (("posit8_es1", "unified"), ("fp8_e4m3", "unified"), ("fp8_e4m3", "planes")):
            P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
            for od in (torch.float16, torch.float32):
                assert torch.equal(msq.qlinear.qlinear(x, P, b, od), msq.qlinear.qlinear(x.to(torch.bfloat16), P, b, od)), (M, N, fo, layout)


def test_vector_ops_fast_rounding_equals_generic(msq):
    """The bfloat16 / nearest kernels round with two integer instructions between exact roundings at the loads and the store
    (csrc/msq_vec.hip Qmid16 / Qin16 / Qout16, v_rcp_f32 for 1 / phi): bit for bit what the run-time-parameter kernels (element codec
    at every step, IEEE division; MSQ_VEC_GENERIC=1) give, on every bfloat16 pattern, on fp32 values off the grid, NaNs with low
    payloads, subnormals, zeros of both signs."""
    specs = msq.specs.finalize_mx_specs({"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32,
                                         "bfloat": 16, "custom_cuda": True})
    g = torch.Generator().manual_seed(77)
    allbf = (torch.arange(65536, dtype=torch.int64) << 16).to(torch.int32).view(torch.float32)
    rnd = torch.randint(-2**31, 2**31 - 1, (1 << 20,), generator=g, dtype=torch.int64).to(torch.int32).view(torch.float32)
    near = torch.randn(1 << 20, generator=g) * torch.tensor([1e-3, 1.0, 30.0, 90.0]).repeat(1 << 18)
    special = torch.tensor([0x7F800001, 0x7FFFFFFF, 0xFF800001, 0x7FC00000, 0x80000000, 0x00000001, 0x80000001, 0x00007FFF, 0x80007FFF,
                            0x80008000, 0x7F7FFFFF, 0xFF7FFFFF, 0x7F7F8000, 0x7F800000, 0xFF800000], dtype=torch.int64).to(torch.int32).view(torch.float32)
    x = torch.cat([allbf, rnd, near, special]).to(dev())
    x = torch.cat([x, torch.zeros((-x.numel()) % 4096, device=dev())])
    y = x.flip(0).contiguous()

    def same(a, b):
        na, nb = torch.isnan(a), torch.isnan(b)
        return bool(torch.equal(na, nb)) and bool(torch.equal(a.view(torch.int32)[~na], b.view(torch.int32)[~nb]))

    def both(fn):
        os.environ.pop("MSQ_VEC_GENERIC", None)
        fast = fn()
        os.environ["MSQ_VEC_GENERIC"] = "1"
        try:
            gen = fn()
        finally:
            os.environ.pop("MSQ_VEC_GENERIC", None)
        return fast, gen

    for fo in (False, True):
        a, b = both(lambda: msq.vector_ops.gelu(x, mx_specs=specs, first_order_gelu=fo))
        assert same(a, b), ("gelu", fo, int((a.view(torch.int32) != b.view(torch.int32)).sum()))
    a, b = both(lambda: msq.vector_ops.simd_add(x, y, mx_specs=specs))
    assert same(a, b), "add"
    a, b = both(lambda: msq.vector_ops.simd_add(x, 1.5, mx_specs=specs))
    assert same(a, b), "add scalar"
    for H in (512, 4096, 8192):
        fin = x[torch.isfinite(x) & (x.abs() < 1e18)]
        rows = fin[: (fin.numel() // H) * H].reshape(-1, H).clone()
        rows[1::2] = torch.randn(rows[1::2].shape, device=dev()) * 3 + 0.5          # ordinary activations on every other row
        rows[3, :7] = torch.tensor([float("nan"), 0.0, -0.0, 1e-40, -1e-40, 3e38, -3e38], device=dev())
        rows[5] = 0.0                                                                # zero variance: 1 / sqrt(eps)
        w = torch.randn(H, device=dev()); w[:4] = torch.tensor([0.0, -0.0, 1e-39, float("nan")], device=dev())
        bias = torch.randn(H, device=dev()); bias[:3] = torch.tensor([-0.0, 0.0, -1e-41], device=dev())
        for eps in (1e-12, 1e-5, 0.0):
            a, b = both(lambda: msq.vector_ops.layer_norm(rows, w, bias, eps, specs))
            assert same(a, b), ("layernorm", H, eps, int((a.view(torch.int32) != b.view(torch.int32)).sum()))


@pytest.mark.parametrize("dn", ["float32", "float16", "bfloat16"])
def test_kv_group_quant_division_free_path_is_exact(msq, O, dn):
    """The group quantiser multiplies by v_rcp_f32(scale) and re-does, with the IEEE division, every chunk in which a value lands
    within levels * 2^-21 of a rounding boundary (csrc/msq_kv.hip kv_codec_n): identical to the oracle on data BUILT to sit on the
    boundaries (x = mn + scale * (k + 1/2) * (1 +- a few ulp)), on constant / NaN / Inf groups, for every kernel form (groups of 32,
    128 and whole-token 4096 along head.dim; 32, 64, 128 tokens of a channel) and 1 ... 8 (and 16) bits."""
    dt = getattr(torch, dn)
    g = torch.Generator().manual_seed(5)
    B, H, S, D = 1, 32, 256, 128
    for bits in (1, 2, 3, 4, 8, 16):
        L = float(2 ** bits - 1)
        base = torch.randn(B, H, S, D, generator=g) * 0.7
        # boundary data: per element k + 1/2 (k < levels) scaled into [a, a + L * sc], nudged by -2 ... 2 ulp
        a = torch.randn(B, H, S, 1, generator=g); sc = torch.rand(B, H, S, 1, generator=g) * 0.3 + 0.01
        k = torch.randint(0, int(min(L, 1 << 20)), (B, H, S, D), generator=g).float()
        xb = a + sc * (k + 0.5)
        xb = (xb.view(torch.int32) + torch.randint(-2, 3, xb.shape, generator=g, dtype=torch.int32)).view(torch.float32)
        xb[..., 0] = a[..., 0]; xb[..., 1] = (a + sc * L)[..., 0]                       # the group's min and max (groups along head.dim)
        x = torch.where(torch.rand(B, H, S, 1, generator=g) < 0.5, xb, base)
        x[0, 0, 0, :] = 1.25                                                            # constant group: 0 / 0
        x[0, 1, 1, 5] = float("nan"); x[0, 2, 2, 9] = float("inf"); x[0, 3, 3, 11] = -float("inf")
        x[0, 4, 4, :] = 0.0; x[0, 5, 5, :64] = 1e-38; x[0, 5, 5, 64:] = 3e-38           # scale in the subnormal range
        x = x.to(dt)
        xd = x.to(dev())
        xf = x.float().numpy()
        for along, gs in ((False, 32), (False, 128), (False, 4096), (False, 1024), (True, 32), (True, 64), (True, 128), (True, 8)):
            f = (msq.kvcache.fake_groupwise_channel_asymmetric_quantization_new if along else msq.kvcache.fake_groupwise_token_asymmetric_quantization)
            y = f(xd, bits, gs).float().cpu().numpy()
            ref = O.kv_group_quant(xf, bits, gs, along, dn)
            same = (y == ref) | (np.isnan(y) & np.isnan(ref))
            assert same.all(), (dn, bits, along, gs, int((~same).sum()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_mx_act_pack_vec_equals_block_kernel(msq, dt):
    """The activation packer of the MX path (one lane per eight values, csrc/msq_mx.hip k_mx_pack_a8_vec) writes the codes, scale
    bytes and status flags of the lane-per-block kernel (MSQ_MX_PACK_BLOCK=1), which the oracle tests pin: random rows, huge and
    tiny blocks, zeros, fp32 subnormals with and without flushing, Inf / NaN blocks."""
    from msq._lib import lib, ptr, current_stream
    torch.manual_seed(61)
    M, K = 333, 1024
    x = torch.randn(M, K, device=dev()) * torch.exp(torch.randn(M, 1, device=dev()) * 4)
    x[0] = 0.0; x[1, :32] = 1e-41; x[2, 32:64] = 3e38; x[3, 5] = float("inf"); x[4, 70] = float("nan"); x[5, :64] = -0.0
    x[6, :32] = torch.linspace(-500, 500, 32, device=dev()); x[7, 64:96] = 2.0 ** -130
    x = x.to(dt).contiguous()
    fn = lib().msq_mx_pack_a8_bf16 if dt == torch.bfloat16 else lib().msq_mx_pack_a8

    def run(flush):
        codes = torch.zeros(M, K, dtype=torch.uint8, device=dev()); scales = torch.zeros(M, K // 32, dtype=torch.uint8, device=dev())
        st = torch.zeros(1, dtype=torch.int32, device=dev())
        assert fn(ptr(x), ptr(codes), ptr(scales), ptr(st), M, K, flush, current_stream(dev())) == 0
        torch.cuda.synchronize()
        return codes, scales, int(st.item())

    for flush in (0, 1):
        os.environ.pop("MSQ_MX_PACK_BLOCK", None)
        a = run(flush)
        os.environ["MSQ_MX_PACK_BLOCK"] = "1"
        try:
            b = run(flush)
        finally:
            os.environ.pop("MSQ_MX_PACK_BLOCK", None)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2], (dt, flush)


def test_harness_benchmark_per_token_latency_and_ppl(msq):
    """harness.benchmark (llm/opt.py:332-376): one token at a time with the KV cache, median latency, PPL with check=True -- on a
    packed tiny Llama the per-token PPL equals the PPL of one full forward over the same tokens (same logits up to fp16 attention
    order), and the fused projections keep working with a growing cache."""
    from msq.harness.benchmark import benchmark
    from msq.harness.evalppl import LLAMA_FUSE, pack_layers, quantize_layers_nearest
    torch.manual_seed(3)
    m = _tiny_llama(torch.float16).to(dev())
    quantize_layers_nearest(m.model.layers, dev(), dict(inlier_elem_format="fp4_e2m1", outlier_elem_format="fp8_e4m3", axes=[-1], block_size=32))
    n, kept = pack_layers(m.model.layers, fuse=LLAMA_FUSE)
    assert n > 0 and kept == 0
    ids = torch.randint(0, m.config.vocab_size, (1, 24), generator=torch.Generator().manual_seed(4))
    lines = []
    r = benchmark(m, ids, check=True, log=lambda *a: lines.append(a))
    assert len(r["times"]) == 24 and r["median"] > 0 and lines[0] == ('Benchmarking ...',) and lines[-2][0] == 'Median:' and lines[-1][0] == 'PPL:'
    with torch.no_grad():
        lg = m(ids.to(dev())).logits[0, :-1].float()
    ref = torch.exp(torch.nn.functional.cross_entropy(lg, ids[0, 1:].to(dev()))).item()
    assert abs(r["ppl"] - ref) / ref < 2e-2, (r["ppl"], ref)


@pytest.mark.parametrize("N,K", [(8256, 512), (12288, 1024), (22016, 704), (16448, 11008 // 64 * 64 // 43)])
def test_wide_projection_decode_kernel(msq, N, K):
    """k_qgemv_u (unified layouts, more than 128 strips, one launch: sixteen-wave blocks up to 256 strips, eight-wave blocks two per
    CU above; K-runs of any length, activation prefetch from two rows on, hand-over summed by all waves): equal to the dense product of
    the unpacked weight with the bf16 activations in fp32 within the accumulation-order tolerance, bit-identical run to run, for one
    and two row groups, every output dtype, with and without bias."""
    K = max(64, K // 64 * 64)
    torch.manual_seed(N + K)
    W = _weights(N, K, 7).to(dev())
    for fo in ("fp8_e4m3", "posit8_es1"):
        P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        Wd = msq.qlinear.unpack_weight(P, torch.float32)
        for M in (1, 2, 16, 17, 32, 33, 36, 48):                       # 1, 2 and 4 row groups (four: up to 36 / 48 rows by width)
            x = torch.randn(M, K, device=dev()).to(torch.bfloat16)
            b = torch.randn(N, device=dev())
            ref = x.float() @ Wd.t()
            tol = 2e-5 * (x.float().abs() @ Wd.abs().t()) + 1e-6
            for bias in (None, b):
                y = msq.qlinear.qlinear(x, P, bias, torch.float32)
                r = ref if bias is None else ref + bias
                assert bool(((y - r).abs() <= tol).all()), (N, K, fo, M, float((y - r).abs().max()))
                assert torch.equal(y, msq.qlinear.qlinear(x, P, bias, torch.float32))
            y16 = msq.qlinear.qlinear(x, P, b, torch.float16)
            ybf = msq.qlinear.qlinear(x, P, b, torch.bfloat16)
            y32 = msq.qlinear.qlinear(x, P, b, torch.float32)
            assert torch.equal(y16, y32.half()) and torch.equal(ybf, y32.to(torch.bfloat16))


def test_fp16_activation_cast_and_mx_pack(msq):
    """An fp16 model at prefill sizes: msq_cast_f16_bf16 == Tensor.to(bfloat16) on every fp16 pattern (NaN payloads aside) and ragged
    lengths; the MX activation packer reads fp16 directly with the codes and scales of packing x.float()."""
    from msq._lib import lib, ptr, current_stream
    allh = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(torch.float16).to(dev())
    for n in (65536, 65531, 7, 8, 1):
        x = allh[:n].contiguous()
        y = torch.empty(n, dtype=torch.bfloat16, device=dev())
        assert lib().msq_cast_f16_bf16(ptr(x), ptr(y), n, current_stream(dev())) == 0
        ref = x.to(torch.bfloat16)
        nan = torch.isnan(ref)
        assert torch.equal(torch.isnan(y), nan) and torch.equal(y.view(torch.int16)[~nan], ref.view(torch.int16)[~nan])
    torch.manual_seed(8)
    x = (torch.randn(300, 1024, device=dev()) * torch.exp(torch.randn(300, 1, device=dev()) * 3)).half()
    x[0] = 0; x[1, :32] = 6e-8; x[2, 5] = 65504.0
    c16, s16 = msq.qlinear.mx_pack_act(x)
    c32, s32 = msq.qlinear.mx_pack_act(x.float())
    assert torch.equal(c16, c32) and torch.equal(s16, s32)
    # and through the Linear: an fp16 activation at M = 300 gives what its bf16 cast gives
    W = _weights(512, 1024, 4).to(dev())
    P = msq.qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 32, layout="unified")
    assert torch.equal(msq.qlinear.qlinear(x, P, None, torch.float16), msq.qlinear.qlinear(x.to(torch.bfloat16), P, None, torch.float16))


def test_mx_decode_wide_projection_eight_wave_blocks(msq):
    """MX decode kernels with the 24- / 32-byte weight operands on more than 256 strips (fused gate / up: 344) take eight-wave blocks,
    two per CU: same tolerance against the exact product of the quantised operands as the other MX kernels, bit-identical run to run."""
    torch.manual_seed(21)
    N, K = 16640, 512                                                   # 260 strips
    W = _weights(N, K, 3).to(dev())
    W8 = msq.quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    bias = torch.randn(N, device=dev())
    for P, Wd in ((msq.qlinear.mx_pack_values(W8), W8.double()),
                  (msq.qlinear.mx_pack_weight(W, w_fmt="e3m2"), msq.mx_ops._quantize_mx(W, 8, "fp6_e3m2", axes=[-1], block_size=32).double())):
        for M in (1, 5, 16):
            x = torch.randn(M, K, device=dev())
            xq = msq.mx_ops._quantize_mx(x, 8, "fp8_e4m3", axes=[-1], block_size=32).double()
            r = xq @ Wd.t() + bias.double()
            y = msq.qlinear.qlinear_mx_w4a8(x, P, bias, torch.float32)
            assert bool(((y.double() - r).abs() <= 2.0 ** -11 * (xq.abs() @ Wd.abs().t()) + 1e-6).all()), (P.w_fmt, M)
            assert torch.equal(y, msq.qlinear.qlinear_mx_w4a8(x, P, bias, torch.float32))
