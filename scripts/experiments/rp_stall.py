"""Debugging aid (round 5): the row-parallel leg of the default bench line measured 1.9-5.0 ms per step where the GPU timeline shows 0.75 ms of
work per step and ONE ~30 ms hole per 20 calls.  This replays the leg alone (RCCL group of one rank) and samples the main thread's stack
every 2 ms: whatever frame the host sits in while no kernel is queued is printed with its dwell time."""
import os, sys, threading, time, traceback, collections
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
import bench, msq
from msq import qlinear
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
W = bench.synth_weight(8192, 28672, dev, seed=100)
P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", "posit8_es1", 2, 32, layout="unified"); del W
rp = qlinear.RowParallelQuantLinear(qlinear.QuantLinear.from_packed(P, None, out_dtype=torch.bfloat16), 1, 0, None, comm="rs_ag", chunks=int(os.environ.get("CHUNKS", 1)),
                                    reduce_dtype=torch.bfloat16, single_rank_collectives=True)
X = torch.randn(2048, 28672, device=dev).to(torch.bfloat16)
main_id = threading.main_thread().ident
samples = []
stop = False
def sampler():
    while not stop:
        fr = sys._current_frames().get(main_id)
        if fr is not None:
            samples.append((time.perf_counter(), "%s:%d %s" % (os.path.basename(fr.f_code.co_filename), fr.f_lineno, fr.f_code.co_name)))
        time.sleep(0.002)
if os.environ.get("SAMPLE", "1") == "1":
    th = threading.Thread(target=sampler, daemon=True); th.start()
import gc
if os.environ.get("NOGC"): gc.disable()
for rep in range(3):
    t0 = time.perf_counter()
    r = bench.rowparallel_measure(rp, X, 2048, dev, True, 1, "rs_ag")
    print("rep", rep, "step %.3f gemm %.3f comm %.3f ms   (host %.1f ms)" % (r["step_ms"], r["gemm_ms"], r["comm_ms"], (time.perf_counter() - t0) * 1e3), flush=True)
stop = True
# dwell: consecutive samples in the same frame
run = []; last = None; t_start = None
for t, f in samples:
    if f != last:
        if last is not None and t - t_start > 0.008: run.append((t - t_start, last))
        last, t_start = f, t
run.sort(reverse=True)
for d, f in run[:12]: print("%.1f ms in %s" % (d * 1e3, f))
dist.destroy_process_group()
