#!/usr/bin/env python3
"""Launch one side kernel a few times (for rocprofv3 --pmc passes).  Usage: python3 scripts/experiments/one_kernel.py layernorm|gelu|kv16|kv32|act0|act1|pack_posit|lowp16|lowp16_h|lowpbf_h|kvmx_keys|kvmx_values|rms_pack|silu_pack"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import kvcache, qlinear, vector_ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
what = sys.argv[1]
sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32, "bfloat": 16, "custom_cuda": True})
X = torch.randn(2048, 4096, device=dev); w = torch.randn(4096, device=dev); b = torch.randn(4096, device=dev)
C16 = torch.randn(1, 32, 4096, 128, device=dev).half(); C32 = C16.float()
W = torch.randn(16384, 4096, device=dev) * 0.02
W16 = W.half()
sp8 = msq.specs.finalize_mx_specs({"w_elem_format": "fp4_e2m1", "a_elem_format": "fp8_e4m3", "scale_bits": 8, "block_size": 32, "bfloat": 16, "custom_cuda": True})
GU = torch.randn(2048, 22016, device=dev) if what == "silu_pack" else None
fn = None if what.startswith(("sk", "gemm3_")) else {"layernorm": lambda: vector_ops.layer_norm(X, w, b, 1e-12, sp), "gelu": lambda: vector_ops.gelu(X, mx_specs=sp),
      "kv16": lambda: kvcache.fake_groupwise_token_asymmetric_quantization(C16, 2, 4096),
      "kv32": lambda: kvcache.fake_groupwise_token_asymmetric_quantization(C32, 2, 4096),
      "act0": lambda: qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", 2, 32, "nearest", False, 0),
      "act1": lambda: qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", 5, 32, "nearest", False, 1),
      "lowp16": lambda: msq.quant.outlier_fakequant(W16, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32),
      "lowp16_h": lambda: msq.quant.outlier_fakequant(W16, 8, 8, "int2", "fp4", 2, 0, 16),               # the harness call (llm/llama.py:229-253)
      "lowpbf_h": lambda: msq.quant.outlier_fakequant(W.to(torch.bfloat16), 8, 8, "int2", "fp4", 2, 0, 16),
      "pack_posit": lambda: qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified"), "gemv_gateup": None,
      # round 5: KV-cache MX-FP8 keys / values in fp16, the fused producers
      "kvmx_keys": lambda: kvcache.mx_quantize_keys(C16, "fp8_e4m3", 32), "kvmx_values": lambda: kvcache.mx_quantize_values(C16, "fp8_e4m3", 32),
      "rms_pack": lambda: vector_ops.rms_norm_mx_pack(X, w, None, 1e-6, sp8), "silu_pack": lambda: vector_ops.silu_mul(GU[:, :11008], GU[:, 11008:], sp8, pack=True)}[what]
if what in ("sk64", "sk128", "sk128f2", "gemm3_128"):     # round 6: k_qgemm_sk on the headline weight (ten launches on ONE packed weight: Infinity-Cache warm, as configs.m_sweep's `ms`)
    Mx = 64 if what == "sk64" else 128
    if what == "sk128f2":
        os.environ["MSQ_GEMM_SK"] = "2"
    if what == "sk128":
        os.environ["MSQ_GEMM_SK"] = "1"
    if what == "gemm3_128":
        os.environ["MSQ_GEMM_SK"] = "0"
    Wh = torch.randn(16384, 4096, device=dev) * 0.02
    Ph = qlinear.pack_weight(Wh, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    xh = torch.randn(Mx, 4096, device=dev).to(torch.bfloat16)
    for _ in range(10):
        qlinear.qlinear(xh, Ph, None, torch.bfloat16)
    torch.cuda.synchronize()
    sys.exit(0)
if what == "gemv_gateup":                                # ten COLD launches of the wide-projection decode kernel (distinct weight copies)
    Wg = torch.randn(22016, 4096, device=dev) * 0.02
    P0 = qlinear.pack_weight(Wg, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified")
    c = lambda t: None if t is None else t.clone()
    Ps = [qlinear.PackedWeight(c(P0.inl), c(P0.out), c(P0.scl), P0.N, P0.K, P0.block, P0.in_kind, P0.out_kind, P0.n, P0.k) for _ in range(10)]
    x1 = torch.randn(1, 4096, device=dev).to(torch.bfloat16)
    for P in Ps:
        qlinear.qlinear(x1, P)
    torch.cuda.synchronize()
    sys.exit(0)
for _ in range(10):
    fn()
torch.cuda.synchronize()
