import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import msq
from oracle import oracle as O
from gpu_common import planted, dev
A = planted(40, 256, 29); A[3, :] = 0; A[7, 5] = np.float32(3e38)
e, m, _, mx, _ = msq.formats._get_format_params("fp4_e2m1")
At = torch.from_numpy(A).to(dev())
for tile in (16, 32, 5):
  for rnd in (0, 1, 2):
    y = msq.funcs.quantize_mx_by_tile_func_cuda(At, 8, e, m, mx, tile, 0, False, rnd, python_divisor=True, python_exponent=True).cpu().numpy()
    yo = O.quantize_mx(A, 8, "fp4_e2m1", axis=0, block_size=tile, round=["nearest", "floor", "even"][rnd], plus_eps_defect=True)
    bad = ~((y == yo) | (np.isnan(y) & np.isnan(yo)))
    print(tile, rnd, bad.sum(), [(int(i), int(j), float(A[i, j]), float(y[i, j]), float(yo[i, j]), float(np.abs(A[(i//tile)*tile:(i//tile+1)*tile, j]).max())) for i, j in zip(*np.nonzero(bad))][:6])
