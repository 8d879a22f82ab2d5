"""Second-order weight calibration with MicroScopiQ pruning -- the role of llm/gptq.py (class GPTQ :17,
Hessian accumulation :32-58, the column solver :60-184, free :186-193), written for the GPU:

* the running Hessian 2/n X X^T lives on the device and is updated with one GEMM per calibration batch;
* the column loop of a 128-column block (:113-157: quantise the column with quantize_mx_outlier_hessian, zero the
  `num_outliers` least important entries, feed the error to the columns on the right) is ONE launch of
  ``msq_gptq_block`` (csrc/msq_gptq.hip): error columns in LDS, lazily rebuilt columns with the reference's rounding
  sequence, the block quantiser of the fake-quant kernel, exact radix selection for the pruning step.  The reference
  issues ~150 eager ops per column; round 1 of this build ~25;
* ties of the pruning choice go to the lowest row index (``torch.topk(..., largest=False)`` leaves the choice among
  equal importances unspecified; the CPU, CUDA and HIP implementations all differ).  With that rule fixed the solver
  reproduces the reference's output bit for bit when it is given the same inverse-Hessian factor
  (tests/golden/gptq_exact.npz);
* the block-to-block update (:163) stays a library GEMM.

Quantiser configurations the block kernel does not cover (posit inliers, block 128, a ``groupsize`` with a quantiser whose
``find_params`` is not MXQuantizer's empty one) and grids the device cannot hold resident at once (the kernel's grid barrier
needs every workgroup on a CU: ``msq_gptq_block`` checks that against the device's CU count and refuses otherwise) take the
per-column path below (one fused quantiser launch per column, same tie rule)."""
import math
import time

import torch
import torch.nn as nn

# Arithmetic of the two dense steps around the block kernel (scripts/experiments/gptq_exactness.py measures what they change).
# DEFAULTS = the reference's behaviour: the inverse-Hessian factor is torch's float32 Cholesky / cholesky_inverse / Cholesky
# (llm/gptq.py:98-104), including its error on a Hessian that is not positive-definite in float32
# (`linalg.cholesky: ... not positive-definite`); the block-to-block update (llm/gptq.py:163) is a float32 library GEMM.
# Opt-in: FACTOR_FP64 computes the factor in float64 and rounds it once (the correctly rounded factor: rocSOLVER's float32
# factorisation carries its own rounding noise, different from the CPU LAPACK noise of the reference -- and it accepts Hessians
# that are singular only in float32); UPDATE_FP64 accumulates the update in float64 and rounds once.
FACTOR_FP64 = False
UPDATE_FP64 = False

from ..quant import quantize_mx_outlier_hessian

DEBUG = False


def _as_matrix(layer):
    w = layer.weight.data
    return w.flatten(1) if isinstance(layer, nn.Conv2d) else w


class GPTQ:
    """Same surface as the reference class: ``add_batch`` per calibration batch, ``fasterquant`` once, ``free``."""

    def __init__(self, layer):
        self.layer = layer
        self.dev = layer.weight.device
        self.rows, self.columns = _as_matrix(layer).shape
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0

    # ------------------------------------------------------------------ Hessian
    def _features(self, inp):
        """Calibration input -> [in_features, samples] (llm/gptq.py:36-50)."""
        if isinstance(self.layer, nn.Conv2d):
            cols = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation, padding=self.layer.padding,
                             stride=self.layer.stride)(inp)
            return cols.permute([1, 0, 2]).flatten(1)
        if inp.dim() == 3:
            inp = inp.reshape(-1, inp.shape[-1])
        return inp.t()

    def add_batch(self, inp, out):
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        batch = inp.shape[0]
        feats = self._features(inp).float()
        total = self.nsamples + batch
        self.H.mul_(self.nsamples / total)                    # running mean over all samples seen so far
        self.nsamples = total
        feats = feats * math.sqrt(2.0 / total)
        self.H.addmm_(feats, feats.t())

    # ------------------------------------------------------------------ solver
    def _inverse_factor(self, H, percdamp):
        """Damped H -> upper Cholesky factor of H^-1 (llm/gptq.py:96-103)."""
        idx = torch.arange(self.columns, device=self.dev)
        H[idx, idx] += percdamp * torch.mean(torch.diag(H))
        if FACTOR_FP64:
            Hd = H.double()
            L = torch.linalg.cholesky(Hd)
            return torch.linalg.cholesky(torch.cholesky_inverse(L), upper=True).float()
        L = torch.linalg.cholesky(H)
        return torch.linalg.cholesky(torch.cholesky_inverse(L), upper=True)

    def _quantize_column(self, w, d, qz):
        """One column: fused MicroScopiQ quantiser + pruning of the least important entries.  Returns the new
        column and the (device) number of pruned entries."""
        q, outliers_per_block = quantize_mx_outlier_hessian(
            w.unsqueeze(1).contiguous(), qz.inlier_scale_bits, qz.outlier_scale_bits, qz.inlier_elem_format,
            qz.outlier_elem_format, qz.shared_exp_method, qz.std_dev, qz.axes, qz.block_size, qz.round,
            qz.flush_fp32_subnorms, qz.custom_cuda)
        q = q.flatten()
        n_out = outliers_per_block.sum().to(torch.int16).to(torch.int64)          # llm/gptq.py:147 (int16 as there)
        importance = (q * q) / (d * d)
        order = torch.argsort(importance, stable=True)
        drop = torch.arange(q.numel(), device=q.device) < n_out                    # first n_out of the ordering
        q = q.index_put((order,), torch.where(drop, torch.zeros_like(q), q[order]))
        return q, n_out

    @torch.no_grad()
    def fasterquant(self, blocksize=128, percdamp=.01, groupsize=-1, actorder=False, static_groups=False, verbose=True,
                    hinv=None, per_column=False):
        """`self.quantizer` is an MXQuantizer set by the caller (llm/llama.py:102-113).  ``hinv`` (tests): use this upper
        Cholesky factor of the inverse Hessian instead of computing it; ``per_column`` forces the per-column path.

        ``groupsize`` / ``static_groups`` (llm/gptq.py:81-87,119-127): with a group size the reference re-runs
        ``quantizer.find_params`` on the columns of every group (or, static, swaps in per-group copies made before the
        solve).  ``MXQuantizer.find_params`` is empty (utils/quant.py:429), so for it the groups change nothing and the
        block kernel stays valid; a quantiser whose ``find_params`` does something gets the reference's literal
        per-column sequence."""
        t0 = time.time()
        W = _as_matrix(self.layer).clone().float()
        qz = self.quantizer
        if not qz.ready():
            qz.find_params(W, weight=True)
        if groupsize != -1 and (not isinstance(groupsize, int) or groupsize <= 0):
            raise ValueError("groupsize must be -1 or a positive number of columns")
        if static_groups and groupsize == -1:
            raise ValueError("static_groups needs a groupsize")
        H, self.H = self.H, None
        unused = torch.diag(H) == 0                                                # inputs never seen: freeze at zero
        H[unused, unused] = 1
        W[:, unused] = 0
        groups = None
        if static_groups:                                                          # llm/gptq.py:81-87 (before the reordering, as there)
            import copy
            groups = []
            for i in range(0, self.columns, groupsize):
                g = copy.deepcopy(qz)
                g.find_params(W[:, i:i + groupsize], weight=True)
                groups.append(g)
        from ..quant import MXQuantizer
        grouped = groupsize != -1 and type(qz).find_params is not MXQuantizer.find_params
        perm = None
        if actorder:
            perm = torch.argsort(torch.diag(H), descending=True)
            W, H = W[:, perm], H[perm][:, perm]
        U = self._inverse_factor(H, percdamp) if hinv is None else hinv.to(self.dev).float().contiguous()
        Q = torch.zeros_like(W)
        loss = torch.zeros((), device=self.dev)
        pruned = torch.zeros((), dtype=torch.int64, device=self.dev)
        # the quantiser's per-call NaN check reads a status word back (utils/quant.py:225 asserts per call): inside the
        # column loop that is one host round trip per column; a NaN scale poisons the column, so the check is made
        # once on the result instead
        from .. import quant as _quant
        check_nan, _quant.CHECK_NAN = _quant.CHECK_NAN, False
        try:
            done = False
            if not per_column and not grouped and self._block_kernel_ok(qz, blocksize):
                from .._lib import MSQ_ERR_UNSUPPORTED, MsqError
                W0 = W.clone()
                try:
                    loss, pruned = self._solve_blocks(W, U.contiguous(), Q, qz, blocksize)
                    done = True
                except MsqError as e:            # the grid does not fit this device (partitioned / CU-masked GPU): per-column path
                    if e.rc != MSQ_ERR_UNSUPPORTED:
                        raise
                    W = W0
                    Q.zero_()
            if not done:
                self._solve(W, U, Q, qz, blocksize, loss, pruned,
                            group=(groupsize, groups, perm) if groupsize != -1 else None)
        finally:
            _quant.CHECK_NAN = check_nan
        torch.cuda.synchronize()
        st = getattr(self, "_status", None)
        if st is not None and int(st.item()) & 4:
            from .._lib import MsqError
            raise MsqError("msq_gptq_block: grid barrier timed out -- its workgroups were not all resident (another stream or "
                           "process holds CUs); rerun alone on the device or with per_column=True")
        if check_nan and ((st is not None and int(st.item()) & 1) or bool(torch.isnan(Q).any())):
            raise AssertionError("outlier_val / inlier_val / shared_exp contains NaN values")
        self.error = float(loss.item())
        self.n_pruned = int(pruned.item())
        if verbose:
            print('time %.2f' % (time.time() - t0))
            print('error', self.error)
        if actorder:
            Q = Q[:, torch.argsort(perm)]
        self.layer.weight.data = Q.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)

    def _block_kernel_ok(self, qz, blocksize):
        axes = qz.axes if isinstance(qz.axes, (list, tuple)) else [qz.axes]
        return (blocksize <= 128 and qz.block_size in (8, 16, 32, 64) and [a % 2 for a in axes] == [0]
                and not str(qz.inlier_elem_format).startswith("posit") and self.rows <= 51200
                and qz.shared_exp_method == "max")

    def _solve_blocks(self, W, U, Q, qz, blocksize):
        """One msq_gptq_block launch per column block + one GEMM for the blocks to the right (llm/gptq.py:106-163)."""
        from .._lib import check, current_stream, lib, ptr
        from ..formats import RoundingMode, format_id
        O, K = W.shape
        L = lib()
        loss = torch.zeros((), dtype=torch.float64, device=self.dev)
        pruned = torch.zeros((), dtype=torch.int64, device=self.dev)
        status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        wsb = L.msq_gptq_block_workspace_bytes(O, min(blocksize, K))
        ws = torch.empty(wsb, dtype=torch.uint8, device=self.dev)
        for c0 in range(0, K, blocksize):
            c1 = min(c0 + blocksize, K)
            cols = c1 - c0
            Wt = W[:, c0:c1].t().contiguous()
            Qt, Et = torch.empty_like(Wt), torch.empty_like(Wt)
            check(L.msq_gptq_block(ptr(Wt), U.data_ptr() + (c0 * K + c0) * 4, K, ptr(Qt), ptr(Et), ptr(loss), ptr(pruned),
                                   ptr(status), ptr(ws), wsb, O, cols, int(qz.block_size), format_id(qz.inlier_elem_format),
                                   format_id(qz.outlier_elem_format), int(qz.inlier_scale_bits), int(qz.outlier_scale_bits),
                                   float(qz.std_dev), int(RoundingMode[qz.round]), int(bool(qz.flush_fp32_subnorms)),
                                   current_stream(self.dev)), "msq_gptq_block")
            Q[:, c0:c1] = Qt.t()
            if c1 < K:
                if UPDATE_FP64:
                    W[:, c1:] -= Et.t().double().matmul(U[c0:c1, c1:].double()).float()
                else:
                    W[:, c1:] -= Et.t().matmul(U[c0:c1, c1:])                          # llm/gptq.py:163
        self._status = status
        return loss, pruned

    def _solve(self, W, U, Q, qz, blocksize, loss, pruned, group=None):
        for c0 in range(0, self.columns, blocksize):
            c1 = min(c0 + blocksize, self.columns)
            Wb = W[:, c0:c1].clone()
            Ub = U[c0:c1, c0:c1]
            Eb = torch.zeros_like(Wb)
            for j in range(c1 - c0):
                w, d = Wb[:, j], Ub[j, j]
                if group is not None:                                              # llm/gptq.py:119-127
                    groupsize, groups, perm = group
                    if groups is None:
                        if (c0 + j) % groupsize == 0:
                            qz.find_params(W[:, (c0 + j):(c0 + j + groupsize)], weight=True)
                    else:
                        idx = c0 + j
                        if perm is not None:
                            idx = int(perm[idx])
                        qz = self.quantizer = groups[idx // groupsize]
                q, n_out = self._quantize_column(w, d, qz)
                pruned += n_out
                Q[:, c0 + j] = q
                e = (w - q) / d
                loss += (e * e).sum() / 2                                          # (w - q)^2 / d^2 / 2
                Wb[:, j:] -= torch.outer(e, Ub[j, j:])                             # feed the error to the columns to the right
                Eb[:, j] = e
            W[:, c1:] -= Eb.matmul(U[c0:c1, c1:])                                  # ... and to the blocks to the right

    def free(self):
        self.H = None
        self.Losses = None
        self.Trace = None
        torch.cuda.empty_cache()
