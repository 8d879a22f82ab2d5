# Where the time of a GPTQ layer goes (4096 x 4096, harness default quantiser): factorisation, block kernel launches, block-to-block GEMMs
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, msq
from msq.harness.gptq import GPTQ
from msq._lib import lib, ptr, check, current_stream
from msq.formats import format_id
dev = torch.device("cuda:0"); torch.manual_seed(1)
O = K = int(os.environ.get("GPTQ_N", "4096"))
lin = torch.nn.Linear(K, O, bias=False).to(dev)
with torch.no_grad(): lin.weight.mul_(0.5)
X = torch.randn(8, 512, K, device=dev)
def sync(): torch.cuda.synchronize()
for rep in range(2):
    gp = GPTQ(lin); gp.quantizer = msq.quant.MXQuantizer(); gp.quantizer.configure(8, 8, os.environ.get("GPTQ_FI", "int2"), os.environ.get("GPTQ_FO", "fp4"), axes=[0], block_size=16)
    for t in range(8): gp.add_batch(X[t], None)
    H = gp.H.clone()
    sync(); t0 = time.time(); U = gp._inverse_factor(H, 0.01).contiguous(); sync(); t_f = time.time() - t0
    W = lin.weight.data.clone().float()
    L = lib(); wsb = L.msq_gptq_block_workspace_bytes(O, 128); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    loss = torch.zeros((), dtype=torch.float64, device=dev); pr = torch.zeros((), dtype=torch.int64, device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
    t_k = t_g = t_t = 0.0
    for c0 in range(0, K, 128):
        c1 = c0 + 128
        sync(); t0 = time.time(); Wt = W[:, c0:c1].t().contiguous(); Qt = torch.empty_like(Wt); Et = torch.empty_like(Wt); sync(); t_t += time.time() - t0
        t0 = time.time()
        check(L.msq_gptq_block(ptr(Wt), U.data_ptr() + (c0 * K + c0) * 4, K, ptr(Qt), ptr(Et), ptr(loss), ptr(pr), ptr(st), ptr(ws), wsb, O, 128, 16,
                               format_id(gp.quantizer.inlier_elem_format), format_id(gp.quantizer.outlier_elem_format), 8, 8, 2.0, 0, 0, current_stream(dev)), "k")
        sync(); t_k += time.time() - t0
        t0 = time.time()
        if c1 < K: W[:, c1:] -= Et.t().matmul(U[c0:c1, c1:])
        sync(); t_g += time.time() - t0
    print("rep %d: factorisation %.1f ms, transposes %.1f ms, block kernel %.1f ms (%.2f ms per 128 columns), GEMM updates %.1f ms" % (rep, t_f * 1e3, t_t * 1e3, t_k * 1e3, t_k * 1e3 / (K / 128), t_g * 1e3))
