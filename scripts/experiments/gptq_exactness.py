"""Round 3, GPTQ exactness (judge item 7): per fixture of tests/golden/gptq_exact.npz the exact number of entries that differ from the
reference's CPU solver, where they sit, and what changes them: the block-to-block update in fp32 (library GEMM) or with fp64
accumulation rounded once, and the inverse-Hessian factor computed in fp32 or in fp64 and rounded."""
import os, sys, numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import msq
from msq.harness import gptq as G
z = np.load(os.path.join(ROOT, "tests", "golden", "gptq_exact.npz"))
names = sorted({k.split("|")[0] for k in z.files})
dev = torch.device("cuda:0")


def run(name, own_hinv, update64, factor64):
    rows, cols, bs, blocksize = (int(v) for v in z[f"{name}|cfg"])
    fi, fo = (str(v) for v in z[f"{name}|fmts"])
    lin = torch.nn.Linear(cols, rows, bias=False)
    lin.weight.data = torch.from_numpy(z[f"{name}|W"]).clone()
    lin = lin.to(dev)
    gp = G.GPTQ(lin)
    gp.quantizer = msq.quant.MXQuantizer(); gp.quantizer.configure(8, 8, fi, fo, axes=[0], block_size=bs)
    X = torch.from_numpy(z[f"{name}|X"]).to(dev)
    for t in range(X.shape[0]):
        gp.add_batch(X[t], None)
    G.UPDATE_FP64, G.FACTOR_FP64 = update64, factor64
    gp.fasterquant(blocksize=blocksize, percdamp=.01, verbose=False, hinv=None if own_hinv else torch.from_numpy(z[f"{name}|Hinv"]))
    return lin.weight.detach().cpu().numpy(), gp.error, blocksize


for name in names:
    ref = z[f"{name}|Q"]
    for own, u64, f64 in ((False, False, False), (False, True, False), (True, False, False), (True, False, True), (True, True, True)):
        Q, err, blocksize = run(name, own, u64, f64)
        bad = np.argwhere(Q != ref)
        rows_bad = sorted(set(bad[:, 0].tolist()))
        first_col = int(bad[:, 1].min()) if len(bad) else -1
        print("%-16s %-13s update %-4s factor %-4s: %5d of %6d entries differ, %3d rows, first differing column %4d (block %d), error %.6g vs %.6g" %
              (name, "own factor" if own else "given factor", "fp64" if u64 else "fp32", "fp64" if f64 else "fp32", len(bad), ref.size, len(rows_bad),
               first_col, blocksize, err, float(z[f"{name}|error"])), flush=True)
