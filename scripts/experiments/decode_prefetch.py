#!/usr/bin/env python3
"""Does touching the NEXT projection's packed planes on a side stream (so that they sit in the 256 MB Infinity Cache when its
decode kernel starts) shorten a chain of cold-weight decode launches?  One HIP graph per variant over the four fused projections
of L Llama-2-7B layers (distinct weights, > 1 GB in total)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import msq
from msq import qlinear

dev = torch.device("cuda:0")
torch.manual_seed(0)
fo = "posit8_es1" if "posit" in sys.argv[1:] else "fp8_e4m3"
L = 6
M = 1


def clone(P):
    c = lambda t: None if t is None else t.clone()
    return qlinear.PackedWeight(c(P.inl), c(P.out), c(P.scl), P.N, P.K, P.block, P.in_kind, P.out_kind, P.n, P.k)


base = []
for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
    W = torch.randn(N, K, device=dev) * 0.02
    base.append(qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified"))
    del W
chain = []
for l in range(L):
    for P in base:
        chain.append(clone(P))
xs = {4096: torch.randn(M, 4096, device=dev).to(torch.bfloat16), 11008: torch.randn(M, 11008, device=dev).to(torch.bfloat16)}
tot_bytes = sum(P.nbytes for P in chain)
sink = torch.zeros(len(chain), dtype=torch.int64, device=dev)


def touch(P, i):
    # stand-in for a prefetch kernel: a streaming read of the code plane (the bulk of the bytes)
    v = P.out.view(torch.int64)
    sink[i] = v.sum()


def run_plain():
    for P in chain:
        qlinear.qlinear(xs[P.k], P)


def run_prefetch(side):
    main = torch.cuda.current_stream()
    for i, P in enumerate(chain):
        if i + 1 < len(chain):
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)                       # the touch of weight i + 1 starts when Linear i may start
            with torch.cuda.stream(side):
                touch(chain[i + 1], i + 1)
        qlinear.qlinear(xs[P.k], P)
    main.wait_stream(side)


def timed(fn, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t0 = timed(run_plain)
side = torch.cuda.Stream()
t1 = timed(lambda: run_prefetch(side))
print(f"{fo} M{M}: {L} layers x 4 projections, {tot_bytes/1e6:.0f} MB packed: plain {t0/L:.1f} us per layer ({tot_bytes/t0/1e3:.0f} GB/s), "
      f"with next-weight touch on a side stream {t1/L:.1f} us per layer ({tot_bytes/t1/1e3:.0f} GB/s)")
