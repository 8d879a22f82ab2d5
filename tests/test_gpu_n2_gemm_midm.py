"""Row N2 (fused unpack-dequant-GEMM), the regime between decode and prefill: k_qgemm_sk (csrc/msq_gemm_stream.hip) -- one strip (or strip pair) per
block, K cut over the block's waves, the partial tiles summed in LDS in a fixed order.  The op is number_system/mx/linear.py:91 `F.linear` on
weights whose values are those of utils/quant.py:147-266, at the batch sizes `llm/opt.py:332-376` (`benchmark`) and chunked prefill produce."""
import ctypes

import numpy as np
import pytest
import torch

from gpu_common import dev, weights

pytestmark = pytest.mark.gpu


def _case(O, M, N, K, fo, seed):
    W = weights(N, K, seed)
    X = torch.randn(M, K, generator=torch.Generator().manual_seed(seed + 1)).to(torch.bfloat16)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(seed + 2))
    Wo = O.outlier_fakequant(W.numpy(), 8, 8, "fp4_e2m1", fo, 2, -1, 32)["out"]
    ref = O.linear(X.float().numpy(), Wo, bias.numpy())
    return W, X, bias, ref


@pytest.mark.parametrize("form,M,N,K", [
    (1, 33, 256, 64 * 8), (1, 64, 512, 64 * 16), (1, 50, 256, 64 * 3), (1, 64, 768, 64 * 11), (1, 130, 256, 64 * 9),
    (2, 65, 256, 64 * 4), (2, 128, 512, 64 * 12), (2, 100, 256, 64 * 5), (2, 128, 256, 64 * 1), (2, 300, 512, 64 * 7),
    (3, 129, 256, 64 * 2), (3, 256, 512, 64 * 8), (3, 200, 256, 64 * 6), (3, 512, 768, 64 * 4), (3, 700, 2048, 64 * 10),
    (4, 65, 256, 64 * 8), (4, 128, 512, 64 * 19), (4, 100, 256, 64 * 3), (4, 128, 256, 64 * 1), (4, 300, 2048, 64 * 9),
    (5, 64, 256, 64 * 4), (5, 200, 768, 64 * 12), (6, 129, 256, 64 * 4), (6, 256, 512, 64 * 8), (6, 200, 256, 64 * 12), (6, 700, 2048, 64 * 20)])
@pytest.mark.parametrize("fo", ["posit8_es1", "fp8_e4m3"])
def test_forced_forms_against_the_oracle(msq, O, form, M, N, K, fo, monkeypatch):
    """Each form FORCED (MSQ_GEMM_SK = 1: 64-row strip, eight waves; 2: 128-row strip, four waves; 3: 128 x 128 block, two k-groups of two
    waves; 4: 128-row strip, eight waves with one activation buffer each and a two-phase reduction; 5: 64 x 128 blocks; 6: 128 x 128 blocks with
    EIGHT waves, two activation buffers per k-group, one barrier per tile that retires the reads of a buffer and publishes the next tile) against O.linear on the ORACLE's fake-quant weight: K of 1 ... 16 tiles (fewer tiles than waves, tile counts that are not multiples
    of the ring of three, an odd count per wave), ragged M (the last row block clamps its loads and masks its stores), several row blocks,
    panel counts that are and are not multiples of 8 (both block orders); bias, float32 / bfloat16 / float16 outputs (one sum, rounded
    once); 20 launches bit-identical."""
    W, X, bias, ref = _case(O, M, N, K, fo, 100 * form + M)
    P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
    Xd, bd = X.to(dev()), bias.to(dev())
    monkeypatch.setenv("MSQ_GEMM_SK", str(form))
    assert msq._lib.lib().msq_qlinear_kernel_choice(M, N, K, P.out_kind, -1) == 5
    y = msq.qlinear.qlinear(Xd, P, bd, torch.float32)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    assert torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.bfloat16), y.to(torch.bfloat16))
    assert torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.float16), y.to(torch.float16))
    y0 = msq.qlinear.qlinear(Xd, P, None, torch.float32)
    assert np.abs(y0.cpu().numpy() - (ref - bias.numpy()[None, :])).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    for _ in range(20):
        assert torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.float32), y)
    monkeypatch.setenv("MSQ_GEMM_SK", "0")
    a = msq.qlinear.qlinear(Xd, P, bd, torch.float32)               # the other kernels' sum: another fp32 order, within rounding
    assert (a - y).abs().max().item() <= 2e-5 * np.abs(ref).max()
    monkeypatch.delenv("MSQ_GEMM_SK")


@pytest.mark.parametrize("M", [33, 48, 64])
def test_default_rule_takes_the_kernel_on_wide_projections(msq, O, M):
    """The library's own rule (no switch) on a 12288-wide projection (the fused q/k/v of Llama-2-7B: 192 strips) with a short K: kernel choice 5
    (MSQ_KERNEL_STREAMK), no workspace needed, result against the oracle.  M <= 32 keeps the decode kernels, M > 64 the split-K GEMM (two 64-row blocks per strip on the 4096 x 4096 class up to 128)."""
    N, K = 12288, 256
    L = msq._lib.lib()
    assert L.msq_qlinear_kernel_choice(M, N, K, 6, -1) == 5
    assert L.msq_qlinear_kernel_choice(32, N, K, 6, -1) == 0 and L.msq_qlinear_kernel_choice(65, N, K, 6, -1) != 5
    assert L.msq_qlinear_kernel_choice(64, 4096, 11008, 6, -1) != 5 and L.msq_qlinear_kernel_choice(128, 4096, 4096, 6, -1) == 5
    buf = ctypes.create_string_buffer(128)
    assert L.msq_qlinear_kernel_name(M, N, K, 6, -1, 2, buf, 128) == 0 and buf.value.decode().startswith("k_qgemm_sk<6, uint16_t")
    W, X, bias, ref = _case(O, M, N, K, "posit8_es1", 7 + M)
    P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    y = msq.qlinear.qlinear(X.to(dev()), P, bias.to(dev()), torch.float32)
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


def test_tuning_switch_through_the_abi(msq):
    L = msq._lib.lib()
    assert L.msq_set_tuning(b"MSQ_GEMM_SK", 0) == 0
    assert L.msq_qlinear_kernel_choice(128, 16384, 4096, 6, -1) != 5
    assert L.msq_set_tuning(b"MSQ_GEMM_SK", 2) == 0
    assert L.msq_qlinear_kernel_choice(128, 16384, 4096, 6, -1) == 5
    assert L.msq_set_tuning(b"MSQ_GEMM_SK", -(2 ** 31)) == 0      # INT_MIN: back to the environment / the rule


@pytest.mark.parametrize("form,M,N,K", [(1, 64, 2048, 64 * 19), (2, 128, 2048, 64 * 19), (4, 128, 2048, 64 * 19), (4, 128, 512, 64 * 19), (3, 256, 2048, 64 * 18),
                                           (6, 256, 2048, 64 * 20)])
def test_forms_repeat_bit_for_bit_beside_a_copy_stream(msq, O, form, M, N, K):
    """The activation tiles reach LDS by LDS-DMA; their reads are ordered by the issuing wave's counted vmcnt AND a block barrier (nothing else
    orders them -- the first build of the private-ring forms read right behind the wave's own vmcnt and form 4, one buffer deep, returned wrong
    tiles in about one run of three on this very shape).  200 launches per form beside a second stream that keeps 64 MB copies running
    (changing DMA latencies), wave tile counts that differ inside a block (19 tiles over 8 / 4 waves): every result equals the first, and the
    first equals the oracle."""
    import os
    W, X, bias, ref = _case(O, M, N, K, "posit8_es1", 900 + form)
    P = msq.qlinear.pack_weight(W.to(dev()), 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    Xd, bd = X.to(dev()), bias.to(dev())
    os.environ["MSQ_GEMM_SK"] = str(form)
    try:
        y = msq.qlinear.qlinear(Xd, P, bd, torch.float32)
        assert np.abs(y.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
        side = torch.cuda.Stream()
        a = torch.empty(16 << 20, device=dev()); b = torch.empty(16 << 20, device=dev())
        bad = 0
        for it in range(200):
            if it % 4 == 0:
                with torch.cuda.stream(side):
                    b.copy_(a)
            bad += int(not torch.equal(msq.qlinear.qlinear(Xd, P, bd, torch.float32), y))
        torch.cuda.synchronize()
        assert bad == 0
    finally:
        os.environ.pop("MSQ_GEMM_SK", None)
