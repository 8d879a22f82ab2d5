"""BASELINE config 5 through the HIP path at its real shapes: Llama-2-70B down_proj [8192 x 28672], K split over
G = 8 ranks -> per-rank shard [8192 x 3584], M = 2048 tokens.  One GPU here, so the eight shards are
multiplied one after the other and summed in rank order; the collectives themselves run in a world-size-1 RCCL group
(reduce-scatter + all-gather, chunked) and, with two ranks, over gloo in tests/test_host_logic.py."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N70, K70, G70, M70 = 8192, 28672, 8, 2048


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


@pytest.fixture(scope="module")
def layer70(msq):
    g = torch.Generator(device=dev()).manual_seed(70)
    lin = torch.nn.Linear(K70, N70, bias=True, device=dev())
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N70, K70, generator=g, device=dev()) * 0.02)
        lin.weight[torch.rand(N70, K70, generator=g, device=dev()) < 0.005] *= 16.0
        lin.bias.copy_(torch.randn(N70, generator=g, device=dev()))
    X = torch.randn(M70, K70, generator=g, device=dev()).to(torch.bfloat16)
    return lin, X


def _quantizer(msq, fo="fp8_e4m3"):
    q = msq.quant.MXQuantizer()
    q.configure(8, 8, "fp4_e2m1", fo, axes=[-1], block_size=32)
    return q


@pytest.mark.parametrize("fo", ["fp8_e4m3", "posit8_es1"])
def test_row_parallel_70b_shards_bf16_path(msq, O, layer70, fo):
    """RowParallelQuantLinear.from_linear (bf16 activations, fused dequant-GEMM) on the eight [8192 x 3584] shards at
    M = 2048: (1) every shard's dequantised weight is the K slice of the unsharded fake-quant, bit for bit (masks and
    scales do not see the cut); (2) the rank-ordered sum of the eight partial outputs equals the unsharded layer up to
    fp32 summation order (<= 2e-5 max|y|, the tolerance of the fused GEMM against the oracle); (3) against the oracle's
    fake-quant + double-precision linear on a row / column sample it can finish in seconds."""
    lin, X = layer70
    q = _quantizer(msq, fo)
    R = msq.qlinear.RowParallelQuantLinear
    full = msq.qlinear.QuantLinear.from_linear(lin, q, out_dtype=torch.float32)
    y_full = msq.qlinear.qlinear(X, full._packed(), full.bias, torch.float32)
    Wq_full = full.dequantize()
    assert torch.equal(Wq_full, msq.quant.outlier_fakequant(lin.weight.data, 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"])
    acc = torch.zeros(M70, N70, dtype=torch.float32, device=dev())
    for r in range(G70):
        rp = R.from_linear(lin, q, G70, r)
        k0, k1 = R.shard_bounds(K70, G70, r, 32)
        assert (k1 - k0) == 3584 and rp.shard.in_features == 3584 and rp.shard.out_features == N70
        assert (rp.shard.bias is not None) == (r == 0)
        assert torch.equal(rp.shard.dequantize(), Wq_full[:, k0:k1])
        acc += msq.qlinear.qlinear(X[:, k0:k1].contiguous(), rp.shard._packed(), rp.shard.bias, torch.float32)   # rank r's term
        del rp
    scale = float(y_full.abs().max())
    assert float((acc - y_full).abs().max()) <= 2e-5 * scale, float((acc - y_full).abs().max()) / scale
    # oracle on a sample: 48 rows x 384 output columns (blocks run along K: rows of W are independent)
    rows = torch.arange(0, M70, 43, device=dev())[:48]
    cols = torch.cat([torch.arange(0, 128), torch.arange(4000, 4128), torch.arange(N70 - 128, N70)]).to(dev())
    Wo = O.outlier_fakequant(lin.weight.data[cols].cpu().numpy(), 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"]
    assert (Wo == Wq_full[cols].cpu().numpy()).all()
    yo = O.linear(X[rows].float().cpu().numpy(), Wo, lin.bias.data[cols].cpu().numpy())
    got = acc[rows][:, cols].cpu().numpy()
    assert np.abs(got - yo).max() <= 2e-5 * np.abs(yo).max() + 1e-6, np.abs(got - yo).max() / np.abs(yo).max()


def test_row_parallel_70b_shards_mx_path(msq, O, layer70):
    """The same layer on the MX matrix path (path="mx": e4m3 weight operand x MX-FP8 activations): 3584 = 28 x 128, so
    the cut splits neither an activation 32-block nor a packed 128-k tile.  The rank-ordered sum of the shard outputs
    against the exact product of the operands: |err| <= 2^-11 sum|products| (the scaled MFMA's accumulation)."""
    lin, X = layer70
    q = _quantizer(msq)
    R = msq.qlinear.RowParallelQuantLinear
    Xf = X.float()
    Xq = msq.mx_ops._quantize_mx(Xf, 8, "fp8_e4m3", axes=[-1], block_size=32)
    Wq = msq.quant.outlier_fakequant(lin.weight.data, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
    acc = torch.zeros(M70, N70, dtype=torch.float32, device=dev())
    for r in range(G70):
        rp = R.from_linear(lin, q, G70, r, path="mx")
        k0, k1 = R.shard_bounds(K70, G70, r, 32, 128)
        assert isinstance(rp.shard, msq.qlinear.MXLinearW4A8) and rp.shard.in_features == 3584
        acc += rp.shard(Xf[:, k0:k1].contiguous())
        del rp
    rows = torch.arange(0, M70, 16, device=dev())
    ref = Xq[rows].double() @ Wq.double().t() + lin.bias.data.double()
    bound = 2.0 ** -11 * (Xq[rows].double().abs() @ Wq.double().abs().t()) + 1e-5
    assert bool(((acc[rows].double() - ref).abs() <= bound).all())
    # operands vs the oracle on a sample (bit-exact)
    assert (Xq[:8].cpu().numpy() == O.quantize_mx(Xf[:8].cpu().numpy(), 8, "fp8_e4m3", axis=-1, block_size=32)).all()


def test_row_parallel_collective_path_single_rank_rccl(msq, layer70):
    """The forward's collective code (chunked reduce-scatter + all-gather and the all-reduce variant, async on RCCL's
    stream, GEMM writing straight into the reduced buffer) in a world-size-1 RCCL group on this GPU: the result must
    equal the shard's own output bit for bit, for every chunking, ragged M included."""
    import torch.distributed as dist
    lin, X = layer70
    q = _quantizer(msq)
    R = msq.qlinear.RowParallelQuantLinear
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=dev())
    try:
        k0, k1 = R.shard_bounds(K70, G70, 0, 32)
        base = R.from_linear(lin, q, G70, 0)
        for reduce_dtype in (torch.float32, torch.bfloat16):
            shard = msq.qlinear.QuantLinear.from_packed(base.shard._packed(), base.shard.bias, out_dtype=reduce_dtype)
            for M in (M70, 1000):
                x = X[:M, k0:k1].contiguous()
                whole = msq.qlinear.qlinear(x, shard._packed(), shard.bias, torch.float32)
                for comm, chunks in (("rs_ag", 0), ("rs_ag", 3), ("all_reduce", 2)):
                    rp = R(shard, 1, 0, None, comm=comm, chunks=chunks, reduce_dtype=reduce_dtype, single_rank_collectives=True)
                    assert rp.chunks_for(M) > 1 or (chunks == 0 and M < 1024)
                    y = rp(x)
                    torch.cuda.synchronize()
                    # the same GEMM calls on the same row chunks, no collective (a chunk of 512 rows takes the split-K
                    # schedule, so it is compared with itself, and with the one-call result up to fp32 summation order)
                    want = torch.cat([msq.qlinear.qlinear(x[r0:r1], shard._packed(), shard.bias, reduce_dtype)
                                      for r0, r1 in rp.chunk_bounds(M)])
                    assert y.dtype == reduce_dtype and torch.equal(y, want), (comm, chunks, M, reduce_dtype)
                    tol = (2e-5 if reduce_dtype == torch.float32 else 2.0 ** -7) * float(whole.abs().max())
                    assert float((y.float() - whole).abs().max()) <= tol
    finally:
        dist.destroy_process_group()
