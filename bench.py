#!/usr/bin/env python3
"""bench.py -- fused unpack-dequant-GEMM throughput (BASELINE.json metric) on MI355X.

A "step" is one pass of the hot path over one synthetic batch: Y[M,N] = X[M,K] . Wq^T with the
packed MicroScopiQ W4 weight dequantised inside the MFMA kernel (msq_qlinear_bf16).
Workload (BASELINE.json configs[1], north_star "[B.S, H] x [H, 4H]"): Llama-2-7B shape
H = 4096, M = B.S = 2048 tokens, W = [4H, H] = [16384, 4096], MX-FP4 (e2m1) inliers + 8-bit
outliers (posit8_es1 by default, --outlier fp8_e4m3 for the all-HW-convert variant), block 32
along K, scale bits 8/8, std_dev 2, heavy-tailed synthetic weights (0.5 % of entries x16).

N > 1: the 7B path does not shard ("7B = replicas only"): every rank runs an
independent replica on its own GPU, no data-path collective, value = sum over ranks ("weak").
--workload llama70b_rowparallel shards K over the ranks with one RCCL all-reduce per step.

Prints ONE JSON line (rank 0).  The timed region starts with all inputs resident in HBM.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0      # dense fp8 MFMA peak (MI355X_MICROARCH.md); the fp4 x fp8 scaled MFMA issues at the fp8 rate


def synth_weight(N, K, dev, seed=0):
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, K, generator=g, device=dev) * 0.02
    W[torch.rand(N, K, generator=g, device=dev) < 0.005] *= 16.0
    return W


def cpu_baseline(M, N, K, bs, fi, fo, plain_mx=False):
    """The oracle (a scalar C port of the reference's CPU fake-quant + the dense linear it feeds)
    timed on the host on a bounded sample of the same workload.  Reported, never optimised."""
    import numpy as np
    from oracle import oracle as O
    rng = np.random.RandomState(0)
    rows = N                                     # fake-quant sample: the whole [N, K] weight (~10 s)
    Ws = (rng.randn(rows, K) * 0.02).astype(np.float32)
    Ws[rng.rand(rows, K) < 0.005] *= 16
    t0 = time.perf_counter()
    r = {"out": O.quantize_mx(Ws, 8, "fp4_e2m1", axis=-1, block_size=bs)} if plain_mx else O.outlier_fakequant(Ws, 8, 8, fi, fo, 2, -1, bs)
    t_q = time.perf_counter() - t0
    ms, ns = 512, 1024                           # linear sample: [512, K] x [1024, K]^T (~2-4 s)
    Xs = rng.randn(ms, K).astype(np.float32)
    t0 = time.perf_counter()
    O.linear(Xs, r["out"][:ns])
    t_l = time.perf_counter() - t0
    # the reference itself runs torch F.linear on the host cores (number_system/mx/linear.py:91)
    torch_tf, threads = None, None
    try:
        import torch
        threads = torch.get_num_threads()
        Xt = torch.randn(1024, K); Wt = torch.from_numpy(r["out"])
        torch.nn.functional.linear(Xt, Wt)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(Xt, Wt)
        torch_tf = 3 * 2.0 * 1024 * rows * K / (time.perf_counter() - t0) / 1e12
        del Wt
    except Exception:
        pass
    return {
        "value": 2.0 * ms * ns * K / t_l / 1e12, "unit": "TFLOP/s", "cores": 1, "kind": "port",
        "sample": "oracle dense linear X[%d,%d].Wq[%d,%d]^T (double accumulate, 1 thread) after oracle "
                  "fake-quant of W[%d,%d]" % (ms, K, ns, K, rows, K),
        "fakequant_s_per_weight": t_q * (N / rows),
        "fakequant_sample_s": t_q,
        "torch_cpu_fp32_linear_tflops": torch_tf, "torch_cpu_threads": threads,
        "host_cpus": os.cpu_count(),
    }


def ppl_proxy(dev, fi, fo, bs):
    """PPL-delta stand-in (no WikiText-2 / Llama-2 checkpoint in the image): a tiny random-weight Llama evaluated by the
    harness with the weights fake-quantised in place (the reference's path, llm/llama.py:240-253) and then with the same
    Linears swapped for packed QuantLinear modules (fused dequant-GEMM, bf16 activations).  Reports both perplexities
    and the relative difference; BASELINE's "<= 0.05 at PPL ~5.5" corresponds to a relative 0.9 %."""
    import types
    import torch
    import msq
    from msq.harness import find_layers, llama
    from msq.harness.data_utils import _Enc
    from msq.harness.evalppl import perplexity
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    m = LlamaForCausalLM(LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                     num_key_value_heads=4, vocab_size=512, max_position_embeddings=128)).eval()
    m.seqlen = 64
    tokens = _Enc(torch.randint(0, 512, (1, 64 * 6), generator=torch.Generator().manual_seed(1)))
    qc = dict(inlier_elem_format=fi, outlier_elem_format=fo, axes=[-1], block_size=bs)
    ppl_fake = llama.llama_eval(m, tokens, dev, args=types.SimpleNamespace(nearest=True, use_mx=True), quant_cfg=qc)
    q = msq.quant.MXQuantizer(); q.configure(8, 8, **qc)
    for layer in m.model.layers:
        msq.qlinear.make_quant(layer, {n: q for n in find_layers(layer)})
    ppl_packed = perplexity(m, tokens, dev, 64)
    return {"model": "random-weight Llama, 2 layers, hidden 256, vocab 512 (synthetic tokens)", "ppl_fakequant_dense": ppl_fake,
            "ppl_packed_fused": ppl_packed, "relative_delta": abs(ppl_packed - ppl_fake) / ppl_fake,
            "equivalent_delta_at_ppl_5.5": 5.5 * abs(ppl_packed - ppl_fake) / ppl_fake}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--M", type=int, default=2048)
    ap.add_argument("--H", type=int, default=4096)
    ap.add_argument("--inlier", default="fp4_e2m1")
    ap.add_argument("--outlier", default="posit8_es1")
    ap.add_argument("--block", type=int, default=32)
    ap.add_argument("--workload", default="llama7b_w4_fused_gemm",
                    choices=["llama7b_w4_fused_gemm", "llama7b_w4a8", "llama7b_mx_w4a8", "llama7b_msq_w4a8_mx", "llama70b_rowparallel"])
    ap.add_argument("--layout", default="auto", choices=["planes", "unified", "auto"],
                    help="packed layout: planes = MSQ-T1 (fp4 plane + outlier plane), unified = MSQ-U1 (one e4m3 code per weight)")
    ap.add_argument("--mx", action="store_true",
                    help="llama70b_rowparallel on the MX matrix path (MicroScopiQ e4m3 weight operand x MX-FP8 activations)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from msq import qlinear

    M = args.M
    if args.workload == "llama70b_rowparallel":
        H, N = 8192, 8192                                  # Llama-2-70B down_proj: K = 28672 split over ranks
        K_full = 28672
        if K_full % (world * (128 if args.mx else 64)):
            raise SystemExit("K=28672 must split into 64-multiples (128 with --mx) over the ranks")
        K = K_full // world
        name = "Llama-2-70B W4 row-parallel QuantLinear down_proj [8192 x 28672], K split over %d GPU(s)" % world
    else:
        H = args.H
        N, K = 4 * H, H
        name = ("Llama-2-7B MicroScopiQ W4 (MX-FP4 inliers + %s outliers), fused dequant-GEMM "
                "X[%d,%d] x W[%d,%d]^T" % (args.outlier, M, K, N, K))
    W = synth_weight(N, K, dev, seed=rank)
    w4a8 = args.workload == "llama7b_w4a8"
    mxw4a8 = args.workload == "llama7b_mx_w4a8"
    msqmx = args.workload == "llama7b_msq_w4a8_mx" or (args.workload == "llama70b_rowparallel" and args.mx)
    if msqmx and args.workload == "llama70b_rowparallel":
        # K splits on a multiple of 128: the activation's 32-blocks and the weight's packed tiles stay whole, every
        # shard's operands equal the unsharded ones; one RCCL all-reduce of the bf16 partial outputs per step
        args.outlier = "fp8_e4m3"
        name += " on the MX matrix path (e4m3 weight operand x MX-FP8 activations)"
        from msq import quant
        P = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, args.inlier, args.outlier, 2, -1, args.block)["out"])
        X = torch.randn(M, K, device=dev)
        mxw4a8 = True
    elif msqmx:
        # BASELINE config 3 with the MicroScopiQ weight itself on the MX matrix path: the fake-quant values (MX-FP4
        # inliers + fp8_e4m3 outliers, utils/quant.py:147-266) packed exactly as one e4m3 code per weight + E8M0 scale
        # per 32 k (8.25 bits/weight), MX-FP8 activations; a step = activation pack (fp32 in, one pass) + GEMM
        args.outlier = "fp8_e4m3"
        name = ("Llama-2-7B MicroScopiQ W4A8 on the MX matrix path (MX-FP4 inliers + fp8_e4m3 outliers as one exact e4m3 "
                "operand x MX-FP8 activations, scaled MFMA, no dequantisation), act-pack + GEMM X[%d,%d] x W[%d,%d]^T" % (M, K, N, K))
        from msq import quant
        P = qlinear.mx_pack_values(quant.outlier_fakequant(W, 8, 8, args.inlier, args.outlier, 2, -1, args.block)["out"])
        X = torch.randn(M, K, device=dev)
        mxw4a8 = True
    elif mxw4a8:
        # BASELINE config 3 on the CDNA4 MX matrix path: plain OCP-MX operands (mx_ops.py:332-457, block 32), MX-FP4
        # weights x MX-FP8 (e4m3) activations on v_mfma_scale_f32_16x16x128_f8f6f4; a step = activation pack (fp32 in,
        # one pass) + GEMM
        name = ("Llama-2-7B W4A8 on the MX matrix path (MX-FP4 weights x MX-FP8 activations, scaled MFMA, no dequantisation), "
                "act-pack + GEMM X[%d,%d] x W[%d,%d]^T" % (M, K, N, K))
        args.inlier, args.outlier = "fp4_e2m1 (plain MX)", "none"
        P = qlinear.mx_pack_weight(W)
        X = torch.randn(M, K, device=dev)
    elif w4a8:
        # BASELINE config 3 = MXLinear(w=fp4_e2m1, a=fp8_e4m3, block 32) semantics (mx_ops variant, std_dev 5):
        # a step = activation quantisation of X (fp32 in, one pass) + the fused dequant-GEMM
        name = ("Llama-2-7B W4A8 (MX-FP4 weights + FP8-e4m3 activations, MXLinear semantics), act-quant + fused "
                "dequant-GEMM X[%d,%d] x W[%d,%d]^T" % (M, K, N, K))
        args.outlier = "fp4_e2m1"
        from msq.mx_ops import _quantize_mx_outlier_v1
        P = qlinear.pack_values(_quantize_mx_outlier_v1(W, 8, 8, args.inlier, args.outlier, "max", 5, [1], args.block))
        X = torch.randn(M, K, device=dev)
    else:
        P = qlinear.pack_weight(W, 8, 8, args.inlier, args.outlier, 2, args.block, layout=args.layout)
        X = torch.randn(M, K, device=dev).to(torch.bfloat16)
    del W
    torch.cuda.synchronize()

    def step():
        if mxw4a8:
            y = qlinear.qlinear_mx_w4a8(X, P, None, torch.bfloat16)
            if args.workload == "llama70b_rowparallel" and world > 1:
                dist.all_reduce(y)
            return y
        if w4a8:
            return qlinear.qlinear_w4a8(X, P, None, torch.bfloat16, a_elem_format="fp8_e4m3", a_std_dev=5,
                                        a_block_size=args.block, a_variant=1)
        y = qlinear.qlinear(X, P, None, torch.bfloat16)
        if args.workload == "llama70b_rowparallel" and world > 1:
            dist.all_reduce(y)
        return y

    # Bring the GPU out of its idle power state before the contract's W warm-up steps: the first ~50 launches
    # after an idle phase run at lower clocks (measured: W = 20 -> 1219 TFLOP/s, W = 100 -> 1270 with nothing
    # else changed).  Untimed, reported in config.clock_ramp_launches.
    RAMP = 100
    for _ in range(RAMP):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / args.steps          # HIP events on the launch stream
    if world > 1:
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    flops_step = 2.0 * M * N * K                            # algorithmic: dequant flops not counted
    total_flops = flops_step * args.steps * world
    value = total_flops / wall / 1e12
    achieved = flops_step / (kern_ms * 1e-3) / 1e12
    out = {
        "metric": "fused dequant-GEMM TFLOPS (% MFMA peak) + PPL delta, Llama-7B W4 1xMI355X",
        "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("mxfp8 x e4m3 codes (fp32 accumulate)" if msqmx else "mxfp8 x mxfp4 (fp32 accumulate)") if mxw4a8 else "bf16", "data": "synthetic",
        "config": {"workload": name, "M": M, "N": N, "K": K, "block": args.block, "inlier": args.inlier,
                   "outlier": args.outlier, "packed_bits_per_weight": P.bits_per_element, "clock_ramp_launches": RAMP,
                   "layout": {(1, 2): "planes", (1, 3): "planes", (1, 4): "planes", (0, 4): "bf16", (0, 5): "unified",
                              (0, 6): "unified+ext"}.get((getattr(P, "in_kind", -1), getattr(P, "out_kind", -1)), "mx operand order"),
                   "parallelism": ("replicas x%d" % world) if args.workload != "llama70b_rowparallel"
                   else ("row-parallel K/%d + RCCL all-reduce" % world)},
        "pct_of_mfma_peak": 100.0 * (value / world) / (PEAK_FP8_TFLOPS if mxw4a8 else PEAK_BF16_TFLOPS),
        "ppl_delta": None,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP8_TFLOPS if mxw4a8 else PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / (PEAK_FP8_TFLOPS if mxw4a8 else PEAK_BF16_TFLOPS), "traffic": None,
                     "kernel": "k_mxgemm (+ k_mx_pack)" if mxw4a8 else "k_qgemm", "kernel_ms": kern_ms, "flops_per_launch": flops_step},
    }
    # HBM/fabric bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
    # scripts/profile_gpu.sh, scripts/summarize_profiles.py); they cannot be collected from inside this run.
    tag = {"posit8_es1": "posit", "fp8_e4m3": "fp8"}.get(args.outlier)
    if mxw4a8:
        tag = "msq_w4a8_mx" if msqmx else "mx_w4a8"
    prof = os.path.join(ROOT, "profiles", "r01_%s_summary.json" % tag) if tag else None
    if prof and os.path.exists(prof) and args.workload in ("llama7b_w4_fused_gemm", "llama7b_mx_w4a8", "llama7b_msq_w4a8_mx") and (M, H) == (2048, 4096):
        try:
            pj = json.load(open(prof))
            out["roofline"]["traffic"] = pj.get("traffic_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/r01_%s_summary.json (rocprofv3 --pmc, 2*FETCH_SIZE+WRITE_SIZE)" % tag
            out["roofline"]["algorithmic_bytes"] = pj.get("algorithmic_bytes_per_launch")
        except Exception:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "llama7b_w4_fused_gemm":
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):              # the harness prints progress: keep stdout to ONE JSON line
                out["ppl_proxy"] = ppl_proxy(dev, args.inlier, args.outlier, args.block)
        except Exception as e:                                     # the stand-in must never take the bench line down
            out["ppl_proxy"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(M, N, K, args.block, args.inlier, args.outlier, plain_mx=(mxw4a8 and not msqmx))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
