# wall time of the GPTQ solver (harness/gptq.py) on one Llama-7B sized layer
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq.harness.gptq import GPTQ
dev = torch.device("cuda:0"); torch.manual_seed(0)
for (O, I) in [(1024, 1024), (4096, 4096)]:
    lin = torch.nn.Linear(I, O, bias=False).to(dev)
    gp = GPTQ(lin)
    q = msq.quant.MXQuantizer(); q.configure(8, 8, "int2", "fp4", axes=[0], block_size=16)
    gp.quantizer = q
    for _ in range(2):
        gp.add_batch(torch.randn(1, 2048, I, device=dev), None)
    torch.cuda.synchronize(); t0 = time.time()
    gp.fasterquant(blocksize=128, percdamp=.01, verbose=False)
    torch.cuda.synchronize()
    print(f"fasterquant [{O} x {I}]: {time.time()-t0:.2f} s, pruned {gp.n_pruned}, error {gp.error:.4g}", flush=True)
