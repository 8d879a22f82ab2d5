#!/usr/bin/env python3
"""One default bench.py run (no CPU baseline, no perplexity leg), printed as one short line: for sampling the pool's leases."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
env = dict(os.environ, MSQ_PPL_DISABLE="1")
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True, env=env)
lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not lines:
    print("bench.py failed:", out.stderr[-500:])
    sys.exit(1)
d = json.loads(lines[-1])
c = d.get("configs") or {}
f = lambda k: round(c[k]["frac"], 3) if k in c and c[k].get("frac") is not None else None
print("bench line: %.1f TFLOP/s, %.1f us per step, frac %.3f | configs: w4a8_mx %s, plain fp4 %s, fp6 %s, w4a8_mxlinear %s, decode_cold %s, 70B layer %s"
      % (d["value"], d["ms_per_step"] * 1e3, d["roofline"]["frac"], f("w4a8_mx"), f("w4a8_mx_plain_fp4"), f("w6a8_mx_plain_fp6"), f("w4a8_mxlinear"), f("decode_cold"), f("rowparallel_70b_1gpu")))
