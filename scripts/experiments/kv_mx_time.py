"""KV-cache MX variant on a Llama-2-7B layer cache [1, 32, 4096, 128]: keys (blocks of 32 tokens per channel) and values (blocks of
32 channels per token), fp16 / bf16, the half-precision kernel (one launch) against the upcast route; kernel times from events
around 20 back-to-back calls (the Python wrapper's ~15 us of host time is hidden behind the previous launch)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import kvcache, mx_ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
from msq._lib import lib
for pair4 in (0, 1, 0, 1):
  lib().msq_set_tuning(b"MSQ_MX_LOWP_PAIR4", pair4)
  print("keys kernel:", "k_mx_lowp_pair4 (block row over four waves)" if pair4 else "k_mx_lowp_pair (one lane per block pair)")
  for dt in (torch.float16, torch.bfloat16):
      K = torch.randn(1, 32, 4096, 128, device=dev).to(dt); V = torch.randn(1, 32, 4096, 128, device=dev).to(dt)
      for fmt in ("fp8_e4m3", "fp4_e2m1"):
          r = [t(lambda: kvcache.mx_quantize_keys(K, fmt, 32)), t(lambda: kvcache.mx_quantize_values(V, fmt, 32)),
               t(lambda: mx_ops._quantize_mx(K, 8, fmt, axes=[2], block_size=32, compute_dtype="float32")),
               t(lambda: mx_ops._quantize_mx(V, 8, fmt, axes=[3], block_size=32, compute_dtype="float32"))]
          print("%-8s %-9s keys %.1f us  values %.1f us   | upcast route: keys %.1f us  values %.1f us   (%.1f MB in + out)" %
                (str(dt)[6:], fmt, r[0], r[1], r[2], r[3], 2 * K.numel() * 2 / 1e6), flush=True)
