#!/usr/bin/env python3
"""Timings of every kernel on the hot path at the BASELINE sizes (run on the GPU box; HIP events on the
launch stream, inputs resident in HBM).  Writes a text table to stdout; the committed copy lives in
profiles/r03_kernel_timings.txt.  Usage: python scripts/measure_kernels.py [section ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import msq
from msq import _lib as pkg
from msq import qlinear, quant

dev = torch.device("cuda:0")
torch.manual_seed(0)


import time

RAMP_S = 0.25      # the chip needs a few hundred ms of back-to-back work to leave its idle clocks (round 6: the act quantiser read 27 us in the
                   # first process of a box and 19-20 us in every later one with 10-30 warm-up launches; bench.py ramps by time for the same reason)


def t(fn, n=20, warm=30):
    # the first launches after an idle phase run at lower clocks: warm up by TIME, then by count
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < RAMP_S:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):          # eager calls of 10-30 us kernels are bound by the host: a hiccup there only ever ADDS time -- the best of three loops
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / n
        best = dt if best is None or dt < best else best
    return best


def tg(fn, reps=10, n=20):
    """device time per call without the Python / ctypes floor (~15 us): `reps` calls captured into one HIP graph, the replay timed"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < RAMP_S:
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / reps


def synth(N, K):
    W = torch.randn(N, K, device=dev) * 0.02
    W[torch.rand(N, K, device=dev) < 0.005] *= 16
    return W


def sec_fakequant():
    L = pkg.lib()
    print("# msq_outlier_fakequant, W[16384,4096] f32 (algorithmic bytes = 2 * numel * 4)")
    for (axis, bs, fi, fo) in [(-1, 32, "fp4_e2m1", "fp8_e4m3"), (0, 16, "int2", "fp4"), (-1, 32, "fp4_e2m1", "posit8_es1"),
                               (0, 32, "fp4_e2m1", "fp8_e4m3"), (-1, 16, "int2", "fp4")]:
        A = synth(16384, 4096); out = torch.empty_like(A)
        ax = axis % 2; pre = 16384 if ax == 1 else 1; post = 1 if ax == 1 else 4096; al = A.shape[ax]
        def call():
            pkg.check(L.msq_outlier_fakequant(pkg.ptr(A), pkg.ptr(out), None, None, None, None, None, None, 0, 0, pre, al, post, bs,
                                              pkg.format_id(fi), pkg.format_id(fo), 8, 8, 2.0, 0, 0, 0, pkg.current_stream()))
        ms = t(call)
        # the same tensor in bf16 ("same W in fp32 and bf16"): dtype 2, 2 B read + 2 B written per element
        Ab = A.to(torch.bfloat16); outb = torch.empty_like(Ab)
        wsb = L.msq_outlier_workspace_bytes(pre, al, post, bs, 0)           # (round 6: with it the packed kernels take the 16-bit tensor)
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=dev)
        def callb():
            pkg.check(L.msq_outlier_fakequant(pkg.ptr(Ab), pkg.ptr(outb), None, None, None, None, None, pkg.ptr(ws), wsb, 2, pre, al, post, bs,
                                              pkg.format_id(fi), pkg.format_id(fo), 8, 8, 2.0, 0, 0, 0, pkg.current_stream()))
        mb = t(callb)
        print(f"fakequant axis {axis:2d} bs {bs:2d} {fi:9s} {fo:11s}: {ms*1e3:7.1f} us  {2*A.numel()*4/ms/1e6:6.0f} GB/s | bf16 in/out {mb*1e3:7.1f} us  {2*A.numel()*2/mb/1e6:6.0f} GB/s")


def sec_pack():
    print("# msq_outlier_pack / msq_outlier_unpack, W[16384,4096] (pack: numel*4 + packed bytes; unpack: packed + dense bytes;")
    print("#   pack time includes torch allocations and the status read-back)")
    N, K = 16384, 4096
    W = synth(N, K)
    for fo in ("fp8_e4m3", "posit8_es1"):
        for layout in ("planes", "unified"):
            P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
            ms = t(lambda: qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout), 10)
            mu = t(lambda: qlinear.unpack_weight(P, torch.bfloat16))
            mf = t(lambda: qlinear.unpack_weight(P, torch.float32))
            print(f"{fo:11s} {layout:8s} {P.bits_per_element:5.2f} b/w: pack {ms*1e3:6.0f} us ({(N*K*4+P.nbytes)/ms/1e6:5.0f} GB/s) | "
                  f"unpack->bf16 {mu*1e3:5.0f} us ({(P.nbytes+N*K*2)/mu/1e6:5.0f} GB/s) | unpack->f32 {mf*1e3:5.0f} us ({(P.nbytes+N*K*4)/mf/1e6:5.0f} GB/s)")


def sec_gemm():
    print("# msq_qlinear_bf16 (fused dequant-GEMM), bf16 out; hipBLASLt = torch bf16 matmul on the unpacked weight, same data")
    print("#   (eager calls on ONE weight: rows with M <= 128 are bound by the ~14 us Python / launch floor and stream from the Infinity Cache;")
    print("#    decode sizes with cold weights from HIP graphs: scripts/experiments/decode_cold.py -> profiles/r03_decode_cold*.txt)")
    for (N, K) in [(16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008)]:
        W = synth(N, K)
        for fo in ("fp8_e4m3", "posit8_es1"):
            for layout in ("planes", "unified"):
                P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout=layout)
                Wu = qlinear.unpack_weight(P, torch.bfloat16)
                for M in (1, 16, 64, 128, 256, 512, 2048, 8192):
                    X = torch.randn(M, K, device=dev).to(torch.bfloat16)
                    ms = t(lambda: qlinear.qlinear(X, P))
                    line = f"N{N:5d} K{K:5d} {fo:11s} {layout:8s} M{M:5d}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF  packed stream {P.nbytes/ms/1e6:5.0f} GB/s"
                    if layout == "unified" and M <= 512:      # device time without the Python / launch floor (round 6)
                        line += f" | device (graph) {tg(lambda: qlinear.qlinear(X, P))*1e3:6.1f} us"
                    if layout == "planes" and fo == "fp8_e4m3":
                        ms2 = t(lambda: X @ Wu.t())
                        line += f" | hipBLASLt bf16 {ms2*1e3:7.1f} us {2*M*N*K/ms2/1e9:7.1f} TF"
                    print(line, flush=True)
                del P, Wu


def sec_cfgb():
    print("# reference harness default (cfg B: int2 inliers / fp4 outliers, blocks of 16 along out_features, llm/llama.py:229-237):")
    print("#   fake-quant with the fused kernel, values packed as they are (msq_pack_values), fused GEMM")
    N, K = 16384, 4096
    W = synth(N, K)
    Wq = msq.quant.quantize_mx_outlier_v1(W, 8, 8, "int2", "fp4", "max", 2, [0], 16)
    P = qlinear.pack_values(Wq)
    ms = t(lambda: qlinear.pack_values(Wq), 10)
    print(f"pack_values -> kind {P.out_kind} {P.bits_per_element:.2f} b/w: {ms*1e3:6.0f} us ({(N*K*4+P.nbytes)/ms/1e6:5.0f} GB/s, incl. allocations and status read-back)")
    for M in (1, 16, 2048, 8192):
        X = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ms = t(lambda: qlinear.qlinear(X, P))
        print(f"cfg B N{N} K{K} M{M:5d}: {ms*1e3:7.1f} us {2*M*N*K/ms/1e9:7.1f} TF")


def sec_w4a8():
    print("# msq_act_quant_bf16 / msq_qlinear_w4a8, X[2048,4096] f32 (act-quant algorithmic bytes = numel * 6)")
    M, K, N = 2048, 4096, 16384
    X = torch.randn(M, K, device=dev)
    for variant, sd in ((0, 2), (1, 5)):
        ms = tg(lambda: qlinear.act_quant(X, 8, 8, "fp8_e4m3", "fp8_e4m3", sd, 32, "nearest", False, variant))
        print(f"act_quant variant {variant}: {ms*1e3:6.1f} us  {M*K*6/ms/1e6:5.0f} GB/s (device time from HIP-graph replays; variant 1 = statistics pass + quantiser)")
    W = synth(N, K)
    from msq.mx_ops import _quantize_mx_outlier_v1
    P = qlinear.pack_values(_quantize_mx_outlier_v1(W, 8, 8, "fp4_e2m1", "fp4_e2m1", "max", 5, [1], 32))
    ms = t(lambda: qlinear.qlinear_w4a8(X, P, None, torch.bfloat16, a_std_dev=5, a_variant=1))
    print(f"qlinear_w4a8 (MXLinear semantics) M{M} N{N} K{K}: {ms*1e3:6.1f} us  {2*M*N*K/ms/1e9:6.1f} TF")


def sec_mx(w8=False):
    from msq._lib import lib, ptr, check, current_stream
    if w8:
        print("# MicroScopiQ weights on the MX matrix path (msq_mx_pack_w8 + msq_qlinear_mx_w8a8: fake-quant values fp4_e2m1 + fp8_e4m3")
        print("#   outliers as one exact e4m3 operand, 8.25 bits/weight, x MX-FP8 activations), X f32, bf16 out; flops = 2 M N K")
    else:
        print("# MX-native W4A8 (msq_mx_pack_a8 + msq_qlinear_mx_w4a8: MX-FP4 x MX-FP8 on the scaled MFMA), X f32, bf16 out;")
        print("#   flops = 2 M N K; act pack algorithmic bytes = numel * (4 + 1 + 1/32)")
    fn = lib().msq_qlinear_mx_w8a8 if w8 else lib().msq_qlinear_mx_w4a8
    for (N, K) in [(16384, 4096), (4096, 4096), (11008, 4096), (4096, 11008)]:
        W = synth(N, K)
        if w8:
            W = quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"]
            P = qlinear.mx_pack_values(W)
            tpw = t(lambda: qlinear.mx_pack_values(W), 10)
        else:
            P = qlinear.mx_pack_weight(W)
            tpw = t(lambda: qlinear.mx_pack_weight(W), 10)
        for M in (16, 128, 2048, 8192):
            X = torch.randn(M, K, device=dev)
            xc, xs = qlinear.mx_pack_act(X)
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            wsb = lib().msq_qlinear_mx_w4a8_workspace_bytes(M, N, K); ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
            def gemm():
                check(fn(ptr(xc), ptr(xs), ptr(P.codes), ptr(P.scales), None, ptr(y), 2, M, N, K, ptr(ws), wsb, current_stream(dev)), "gemm")
            tg = t(gemm); tp = t(lambda: qlinear.mx_pack_act(X)); te = t(lambda: qlinear.qlinear_mx_w4a8(X, P))
            print(f"N{N:5d} K{K:5d} M{M:5d}: GEMM {tg*1e3:7.1f} us {2*M*N*K/tg/1e9:7.1f} TF | act pack {tp*1e3:6.1f} us {M*K*(5+1/32)/tp/1e6:5.0f} GB/s | "
                  f"end to end {te*1e3:7.1f} us {2*M*N*K/te/1e9:7.1f} TF", flush=True)
        print(f"N{N:5d} K{K:5d} weight pack (offline): {tpw*1e3:6.0f} us {N*K*(4+P.bits_per_element/8)/tpw/1e6:5.0f} GB/s, {P.bits_per_element:.2f} bits/weight")


def sec_a12():
    """the seven plug-in functions of the reference's native module (cpp/funcs.cpp:218-226), HBM-bound: bytes = read + written"""
    from msq import funcs
    print("# a12 native plug-in kernels, 67.1 M elements f32 (algorithmic bytes = 2 * numel * 4; reduce: numel * 4)")
    A = synth(16384, 4096)
    n = A.numel()
    ms = t(lambda: funcs.quantize_elemwise_func_cuda(A, 9, 8, 3.3895313892515355e38, 0, False, True))
    print(f"k_elemwise_f32 (bf16 rounding, quantize_elemwise_func_cuda): {ms*1e3:7.1f} us {2*n*4/ms/1e6:6.0f} GB/s (incl. the output allocation)")
    ms = t(lambda: funcs.quantize_elemwise_func_cuda(A, 5, 4, 448.0, 0, True, True))
    print(f"k_elemwise_f32 (fp8_e4m3, saturating)                      : {ms*1e3:7.1f} us {2*n*4/ms/1e6:6.0f} GB/s")
    Ah = A.half()
    ms = t(lambda: funcs.quantize_elemwise_func_cuda(Ah, 5, 4, 448.0, 0, True, True))
    print(f"k_elemwise_16<half>                                         : {ms*1e3:7.1f} us {2*n*2/ms/1e6:6.0f} GB/s")
    for axis, tile in ((1, 32), (0, 16), (0, 32)):
        ms = t(lambda: funcs.quantize_mx_by_tile_func_cuda(A, 8, 2, 3, 6.0, tile, axis, False, 0))
        print(f"k_mx_tile_{'inner' if axis == 1 else 'generic'} (quantize_mx_by_tile, fp4, axis {axis}, tile {tile:2d})    : {ms*1e3:7.1f} us {2*n*4/ms/1e6:6.0f} GB/s")
    mv = A.abs().amax(dim=1, keepdim=True).contiguous()
    ms = t(lambda: funcs.quantize_mx_func_cuda(A, 8, 2, 3, 6.0, mv, 1, False, 0))
    print(f"k_mx_maxvals (quantize_mx_func_cuda, axis 1 whole rows)       : {ms*1e3:7.1f} us {2*n*4/ms/1e6:6.0f} GB/s")
    for inner in (32, 1024, 4096):
        B = A.reshape(-1, inner)
        ms = t(lambda: funcs.reduce_sum_inner_dim(B))
        m2 = t(lambda: funcs.reduce_max_inner_dim(B))
        print(f"k_reduce_inner sum / max, inner {inner:5d}                       : {ms*1e3:7.1f} / {m2*1e3:7.1f} us {n*4/ms/1e6:6.0f} / {n*4/m2/1e6:6.0f} GB/s")


def sec_lowp():
    print("# msq_outlier_fakequant computed in the tensor dtype (MSQ_DTYPE_F16_NATIVE / BF16_NATIVE), W[16384,4096]; bytes = 2 * numel * 2")
    A = synth(16384, 4096)
    for dt in (torch.float16, torch.bfloat16):
        Ah = A.to(dt)
        for (axis, bs, fi, fo) in [(0, 16, "int2", "fp4"), (-1, 32, "fp4_e2m1", "fp8_e4m3")]:
            keep = quant.CHECK_NAN; quant.CHECK_NAN = False
            try:
                ms = t(lambda: quant.outlier_fakequant(Ah, 8, 8, fi, fo, 2, axis, bs), 10, 5)
            finally:
                quant.CHECK_NAN = keep
            print(f"{str(dt)[6:]:9s} axis {axis:2d} bs {bs} {fi}/{fo}: {ms*1e3:7.1f} us {2*A.numel()*2/ms/1e6:6.0f} GB/s, {A.numel()/ms/1e6:6.1f} G weights/s (Llama-2-7B: {6.5e9/(A.numel()/ms*1e3):.2f} s)")


def sec_kv():
    from msq import kvcache
    print("# msq_kv_group_quant on a Llama-2-7B layer cache [1, 32, 4096, 128] (bytes = 2 * numel * sizeof; device time from HIP-graph replays)")
    for dt in (torch.float16, torch.float32):
        k = torch.randn(1, 32, 4096, 128, device=dev).to(dt)
        for bits in (2, 4):
            mc = tg(lambda: kvcache.fake_groupwise_channel_asymmetric_quantization_new(k, bits, 32))
            mt = tg(lambda: kvcache.fake_groupwise_token_asymmetric_quantization(k, bits, 32))
            mw = tg(lambda: kvcache.fake_groupwise_token_asymmetric_quantization(k, bits, 4096))
            b = 2 * k.numel() * k.element_size()
            print(f"{str(dt)[6:]:8s} {bits} bit: per-channel g32 {mc*1e3:6.1f} us {b/mc/1e6:5.0f} GB/s | per-token g32 {mt*1e3:6.1f} us {b/mt/1e6:5.0f} GB/s | per-token g4096 {mw*1e3:6.1f} us {b/mw/1e6:5.0f} GB/s")
        mk = tg(lambda: kvcache.mx_quantize_keys(k, "fp8_e4m3", 32)); mv = tg(lambda: kvcache.mx_quantize_values(k, "fp8_e4m3", 32))
        print(f"{str(dt)[6:]:8s} MX-FP8: keys (blocks along tokens) {mk*1e3:6.1f} us | values (blocks along head_dim) {mv*1e3:6.1f} us")
        mk = tg(lambda: kvcache.mx_quantize_keys(k, "fp4_e2m1", 32)); mv = tg(lambda: kvcache.mx_quantize_values(k, "fp4_e2m1", 32))
        print(f"{str(dt)[6:]:8s} MX-FP4: keys (blocks along tokens) {mk*1e3:6.1f} us | values (blocks along head_dim) {mv*1e3:6.1f} us")
        keep, quant.CHECK_NAN = quant.CHECK_NAN, False                 # (the status read-back is a host sync: not inside a graph)
        mk = tg(lambda: kvcache.mx_quantize_keys(k, "fp4_e2m1", 32, outlier_format="fp8_e4m3")); mv = tg(lambda: kvcache.mx_quantize_values(k, "fp4_e2m1", 32, outlier_format="fp8_e4m3"))
        quant.CHECK_NAN = keep
        print(f"{str(dt)[6:]:8s} MicroScopiQ fp4 + fp8 outliers (computed in float32): keys {mk*1e3:6.1f} us | values {mv*1e3:6.1f} us")


def sec_vec():
    from msq import vector_ops
    sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32, "bfloat": 16, "custom_cuda": True})
    print("# bfloat-rounded vector ops, X[2048, 4096] f32 (bytes = 2 * numel * 4; add: 3 * numel * 4; device time from HIP-graph replays)")
    X = torch.randn(2048, 4096, device=dev); w = torch.randn(4096, device=dev); b = torch.randn(4096, device=dev)
    n = X.numel()
    ms = tg(lambda: vector_ops.layer_norm(X, w, b, 1e-12, sp)); print(f"msq_vec_layernorm: {ms*1e3:6.1f} us {2*n*4/ms/1e6:5.0f} GB/s")
    ms = tg(lambda: vector_ops.gelu(X, mx_specs=sp)); print(f"msq_vec_gelu     : {ms*1e3:6.1f} us {2*n*4/ms/1e6:5.0f} GB/s")
    Z = torch.randn(2048, 4096, device=dev)
    ms = tg(lambda: vector_ops.simd_add(X, Z, mx_specs=sp)); print(f"msq_vec_add      : {ms*1e3:6.1f} us {3*n*4/ms/1e6:5.0f} GB/s")


SECTIONS = {"a12": sec_a12, "lowp": sec_lowp, "kv": sec_kv, "vec": sec_vec, "fakequant": sec_fakequant, "pack": sec_pack, "gemm": sec_gemm, "cfgb": sec_cfgb, "w4a8": sec_w4a8, "mx": sec_mx, "mx8": lambda: sec_mx(True)}
if __name__ == "__main__":
    names = sys.argv[1:] or list(SECTIONS)
    print("device:", torch.cuda.get_device_name(0))
    for nm in names:
        SECTIONS[nm]()
