"""GPTQ solver with MicroScopiQ pruning -- the surface of llm/gptq.py (GPTQ :17, add_batch :32-58,
fasterquant :60-184, free :186-193).  The Hessian algebra (Cholesky inverse, error feedback) is plain
torch on the GPU; every column quantisation is ONE fused HIP launch of
``quantize_mx_outlier_hessian`` (utils/quant.py:23-146), followed by the MicroScopiQ step that zeroes
the ``num_outliers`` least-important weights of the column (llm/gptq.py:146-153).

This is the SURVEY.md 8(f1) "next" row in its simplest correct form: one launch per column (the
reference does ~150 eager launches per column).  A column-block-batched kernel is future work."""
import math
import time

import torch
import torch.nn as nn

from ..quant import quantize_mx_outlier_hessian

DEBUG = False


class GPTQ:
    def __init__(self, layer):
        self.layer = layer
        self.dev = self.layer.weight.device
        W = layer.weight.data.clone()
        if isinstance(self.layer, nn.Conv2d):
            W = W.flatten(1)
        self.rows, self.columns = W.shape[0], W.shape[1]
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0

    def add_batch(self, inp, out):
        """H <- running mean of 2 X X^T over calibration batches (llm/gptq.py:32-58)."""
        if len(inp.shape) == 2:
            inp = inp.unsqueeze(0)
        tmp = inp.shape[0]
        if isinstance(self.layer, nn.Linear):
            if len(inp.shape) == 3:
                inp = inp.reshape((-1, inp.shape[-1]))
            inp = inp.t()
        elif isinstance(self.layer, nn.Conv2d):
            unfold = nn.Unfold(self.layer.kernel_size, dilation=self.layer.dilation, padding=self.layer.padding,
                               stride=self.layer.stride)
            inp = unfold(inp).permute([1, 0, 2]).flatten(1)
        self.H *= self.nsamples / (self.nsamples + tmp)
        self.nsamples += tmp
        inp = math.sqrt(2 / self.nsamples) * inp.float()
        self.H += inp.matmul(inp.t())

    @torch.no_grad()
    def fasterquant(self, blocksize=128, percdamp=.01, groupsize=-1, actorder=False, static_groups=False, verbose=True):
        """llm/gptq.py:60-184.  `self.quantizer` is an MXQuantizer (set by the caller, llama.py:102-113)."""
        W = self.layer.weight.data.clone()
        if isinstance(self.layer, nn.Conv2d):
            W = W.flatten(1)
        W = W.float()
        tick = time.time()
        qz = self.quantizer
        if not qz.ready():
            qz.find_params(W, weight=True)
        H = self.H
        del self.H
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        W[:, dead] = 0
        if actorder:
            perm = torch.argsort(torch.diag(H), descending=True)
            W = W[:, perm]
            H = H[perm][:, perm]
            invperm = torch.argsort(perm)
        Losses = torch.zeros_like(W)
        Q = torch.zeros_like(W)
        damp = percdamp * torch.mean(torch.diag(H))
        diag = torch.arange(self.columns, device=self.dev)
        H[diag, diag] += damp
        H = torch.linalg.cholesky(H)
        H = torch.cholesky_inverse(H)
        Hinv = torch.linalg.cholesky(H, upper=True)
        n_pruned = 0
        for i1 in range(0, self.columns, blocksize):
            i2 = min(i1 + blocksize, self.columns)
            count = i2 - i1
            W1 = W[:, i1:i2].clone()
            Q1 = torch.zeros_like(W1)
            Err1 = torch.zeros_like(W1)
            Losses1 = torch.zeros_like(W1)
            Hinv1 = Hinv[i1:i2, i1:i2]
            for i in range(count):
                w = W1[:, i]
                d = Hinv1[i, i]
                q, num_outliers_per_block = quantize_mx_outlier_hessian(
                    w.unsqueeze(1).contiguous(), qz.inlier_scale_bits, qz.outlier_scale_bits, qz.inlier_elem_format,
                    qz.outlier_elem_format, qz.shared_exp_method, qz.std_dev, qz.axes, qz.block_size, qz.round,
                    qz.flush_fp32_subnorms, qz.custom_cuda)
                q = q.flatten()
                importance = (q ** 2) / d ** 2
                num_outliers = int(num_outliers_per_block.sum().to(torch.int16).item())      # gptq.py:147
                if num_outliers > 0:
                    least = torch.topk(importance, num_outliers, largest=False).indices       # gptq.py:150
                    q[least] = 0
                    n_pruned += num_outliers
                Q1[:, i] = q
                Losses1[:, i] = (w - q) ** 2 / d ** 2
                err1 = (w - q) / d
                W1[:, i:] -= err1.unsqueeze(1).matmul(Hinv1[i, i:].unsqueeze(0))
                Err1[:, i] = err1
            Q[:, i1:i2] = Q1
            Losses[:, i1:i2] = Losses1 / 2
            W[:, i2:] -= Err1.matmul(Hinv[i1:i2, i2:])
        torch.cuda.synchronize()
        self.error = torch.sum(Losses).item()
        self.n_pruned = n_pruned
        if verbose:
            print('time %.2f' % (time.time() - tick))
            print('error', self.error)
        if actorder:
            Q = Q[:, invperm]
        self.layer.weight.data = Q.reshape(self.layer.weight.shape).to(self.layer.weight.data.dtype)

    def free(self):
        self.H = None
        self.Losses = None
        self.Trace = None
        torch.cuda.empty_cache()
