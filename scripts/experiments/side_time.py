#!/usr/bin/env python3
"""Device time of the side kernels (bfloat-rounded vector ops, KV group quantiser, activation quantiser) without the Python / ctypes
floor: each call is captured ten times into a HIP graph and the replay is timed.  Usage: python scripts/experiments/side_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import kvcache, qlinear, vector_ops

dev = torch.device("cuda:0")


def tg(fn, reps=10, n=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / reps * 1e3


def main():
    torch.manual_seed(0)
    sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32, "bfloat": 16, "custom_cuda": True})
    X = torch.randn(2048, 4096, device=dev); w = torch.randn(4096, device=dev); b = torch.randn(4096, device=dev)
    by = 2 * X.numel() * 4
    for nm, fn in (("layernorm", lambda: vector_ops.layer_norm(X, w, b, 1e-12, sp)), ("gelu", lambda: vector_ops.gelu(X, mx_specs=sp)),
                   ("gelu first order", lambda: vector_ops.gelu(X, mx_specs=sp, first_order_gelu=True)), ("add", lambda: vector_ops.simd_add(X, X, mx_specs=sp))):
        us = tg(fn)
        print(f"msq_vec {nm:17s} X[2048,4096] f32: {us:6.1f} us  {by/us/1e3:5.0f} GB/s  {by/us/1e3/8000:.2f} of HBM")
    for dt in (torch.float16, torch.float32):
        C = torch.randn(1, 32, 4096, 128, device=dev).to(dt)
        by = 2 * C.numel() * C.element_size()
        for nm, fn in (("per-channel g32 2b", lambda: kvcache.fake_groupwise_channel_asymmetric_quantization_new(C, 2, 32)), ("per-token g32 2b", lambda: kvcache.fake_groupwise_token_asymmetric_quantization(C, 2, 32)),
                       ("per-token g4096 2b", lambda: kvcache.fake_groupwise_token_asymmetric_quantization(C, 2, 4096)), ("per-token g4096 4b", lambda: kvcache.fake_groupwise_token_asymmetric_quantization(C, 4, 4096))):
            try:
                us = tg(fn)
            except Exception as e:      # noqa
                print("kv", nm, "failed:", e); continue
            print(f"msq_kv {str(dt)[6:]:8s} {nm:20s}: {us:6.1f} us  {by/us/1e3:5.0f} GB/s  {by/us/1e3/8000:.2f} of HBM")
    Xa = torch.randn(2048, 4096, device=dev)
    for variant, sd in ((0, 2), (1, 5)):
        us = tg(lambda: qlinear.act_quant(Xa, 8, 8, "fp8_e4m3", "fp8_e4m3", sd, 32, "nearest", False, variant))
        print(f"act_quant variant {variant}: {us:6.1f} us  {Xa.numel()*6/us/1e3:5.0f} GB/s")
    # variant 1 (mx_ops statistics) redoes a column sequentially when its std lies next to a float rounding boundary (about one column
    # in 500 000: ~13 % of [2048, 4096] inputs have one); that chain is the tail of the launch, so the time depends on the data:
    # eight inputs, the one-pass row kernel (k_act_quant_rows) against the statistics + quantiser launches (MSQ_ACT_ROWS=0)
    rows, two = [], []
    for seed in range(8):
        Xs = torch.randn(2048, 4096, device=dev, generator=torch.Generator(device=dev).manual_seed(100 + seed))
        for flag, acc in (("1", rows), ("0", two)):
            os.environ["MSQ_ACT_ROWS"] = flag
            acc.append(tg(lambda: qlinear.act_quant(Xs, 8, 8, "fp8_e4m3", "fp8_e4m3", 5, 32, "nearest", False, 1)))
    os.environ.pop("MSQ_ACT_ROWS", None)
    fmt = lambda v: " ".join(f"{x:5.1f}" for x in v)
    print(f"act_quant variant 1, 8 inputs, one pass (k_act_quant_rows): {fmt(rows)}  median {sorted(rows)[4]:.1f} us  {Xa.numel()*6/sorted(rows)[4]/1e3:5.0f} GB/s")
    print(f"act_quant variant 1, 8 inputs, statistics + quantiser      : {fmt(two)}  median {sorted(two)[4]:.1f} us")

if __name__ == "__main__":
    main()
