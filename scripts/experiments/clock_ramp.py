"""How long a box needs from idle to its steady rate on the headline kernel: chunks of 50 launches for SECONDS s after IDLE s of nothing, rate per
0.25 s window; sclk / power of the card of cuda:0 beside it (bench._gpu_sensors)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, msq
from msq import qlinear
dev = torch.device("cuda:0")
W = bench.synth_weight(16384, 4096, dev, seed=0)
P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified"); del W
X = torch.randn(2048, 4096, device=dev).to(torch.bfloat16)
y = torch.empty(2048, 16384, dtype=torch.bfloat16, device=dev)
step = lambda: qlinear.qlinear(X, P, None, torch.bfloat16, out=y)
for _ in range(20): step()
torch.cuda.synchronize()
fl = 2.0 * 2048 * 16384 * 4096
for idle in (float(os.environ.get("IDLE", 3.0)), 0.05):
    time.sleep(idle)
    evs = [torch.cuda.Event(enable_timing=True)]; evs[0].record()
    t0 = time.perf_counter(); marks = []
    while time.perf_counter() - t0 < float(os.environ.get("SECONDS", 5.0)):
        for _ in range(50): step()
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        if len(evs) % 4 == 0: e.synchronize()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
    t = 0.0; acc = 0.0; n = 0; row = []
    for m in ms:
        t += m; acc += m; n += 1
        if acc >= 250.0:
            row.append("%.2fs %.0f" % (t / 1e3, fl * 50 * n / (acc * 1e-3) / 1e12)); acc = 0.0; n = 0
    print("after %.2f s idle: TFLOP/s per 0.25 s window:" % idle, " | ".join(row), flush=True)
    print("   first five 50-launch chunks:", " ".join("%.0f" % (fl * 50 / (m * 1e-3) / 1e12) for m in ms[:5]), "  sensors:", {k: v for k, v in bench._gpu_sensors().items() if k in ("pp_dpm_sclk", "power1_input")})
