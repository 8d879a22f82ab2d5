#!/usr/bin/env python3
"""BASELINE config 1 as composed: OPT-125M-SHAPED decoder Linears ([768,768] x 4, [3072,768], [768,3072]; 12 heads; two layers are
enough) quantised with fp6_e3m2 inliers (examples/run_mx_fp6.sh:2) by the REFERENCE's RTN arithmetic (llm/opt.py:190-218 calls
utils/quant.py:147-266 `quantize_mx_outlier_v1`), whole-model perplexity with the reference's formula (llm/opt.py:236-249).

Run in the build container only (imports /root/reference from a scratch copy; the GPU box has none):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_opt125m.py

The model's weights are NOT stored (two OPT-125M layers are 57 MB): they are a pure function of numpy's frozen legacy generator
(`fill_weights` below, also imported by the test), and the fixture pins them with per-tensor float64 checksums.  Stored: the
checksums, the tokens, the fp32 perplexity, and per configuration the perplexity, the float64 |w| sum over all decoder Linears and the
first rows of every quantised Linear of layer 0 (bit patterns).  Data only -- no reference source."""
import os
import sys
import warnings

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

CONFIGS = {   # name -> (inlier, outlier, axes, block): blocks along in_features (packable as planes) and the harness's own layout
    "fp6_e3m2_fp8_axm1_bs32": ("fp6_e3m2", "fp8_e4m3", [-1], 32),
    "fp6_e3m2_fp8_ax0_bs16": ("fp6_e3m2", "fp8_e4m3", [0], 16),
}
SEQLEN, VOCAB, ROWS = 64, 256, 4


def opt125m_shaped(layers=2):
    from transformers import OPTConfig, OPTForCausalLM
    cfg = OPTConfig(hidden_size=768, ffn_dim=3072, num_hidden_layers=layers, num_attention_heads=12, vocab_size=VOCAB,
                    max_position_embeddings=128, word_embed_proj_dim=768, do_layer_norm_before=True)
    m = OPTForCausalLM(cfg).eval()
    m.seqlen = SEQLEN
    return m


def fill_weights(model, seed=0):
    """Every parameter from numpy's legacy RandomState (bit-stable across numpy / torch versions), in sorted-name order: matrices
    N(0, 0.02^2) with a heavy tail (1 % of the entries x 8), vectors small; LayerNorm weights near 1."""
    rng = np.random.RandomState(seed)
    sums = {}
    with torch.no_grad():
        for name, p in sorted(model.named_parameters()):
            if p.ndim == 2:
                w = (rng.standard_normal(p.shape) * 0.02).astype(np.float32)
                w[rng.random_sample(p.shape) < 0.01] *= 8.0
            elif "layer_norm" in name and name.endswith("weight"):
                w = (1.0 + 0.05 * rng.standard_normal(p.shape)).astype(np.float32)
            else:
                w = (0.01 * rng.standard_normal(p.shape)).astype(np.float32)
            p.copy_(torch.from_numpy(w))
            sums[name] = float(np.abs(w.astype(np.float64)).sum())
    return sums


def tokens():
    return torch.from_numpy(np.random.RandomState(7).randint(0, VOCAB, size=(1, SEQLEN * 6 + 3)).astype(np.int64))


def main():
    sys.path.insert(0, HERE)
    from make_golden import _import_reference
    scratch, quant, *_ = _import_reference()
    import torch.nn as nn

    def ppl_of(model, ids, seqlen):              # llm/opt.py:236-249
        n = ids.numel() // seqlen
        nlls = []
        with torch.no_grad():
            for i in range(n):
                b = ids[:, i * seqlen:(i + 1) * seqlen]
                lg = model(b).logits
                loss = nn.CrossEntropyLoss()(lg[:, :-1, :].reshape(-1, lg.size(-1)), b[:, 1:].reshape(-1))
                nlls.append(loss.float() * seqlen)
        return torch.exp(torch.stack(nlls).sum() / (n * seqlen)).item()

    def find_linears(mod, name=''):              # utils/modelutils.py:8-15
        if type(mod) is nn.Linear:
            return {name: mod}
        r = {}
        for n1, c in mod.named_children():
            r.update(find_linears(c, name + '.' + n1 if name else n1))
        return r

    d = {}
    m = opt125m_shaped()
    sums = fill_weights(m)
    d["param_names"] = np.array(sorted(sums))
    d["param_abs_sums"] = np.array([sums[k] for k in sorted(sums)], dtype=np.float64)
    ids = tokens()
    d["tokens"] = ids.numpy()
    d["ppl_fp32"] = np.float64(ppl_of(m, ids, SEQLEN))
    import copy
    for cname, (fi, fo, ax, bs) in CONFIGS.items():
        mq = copy.deepcopy(m)
        for layer in mq.model.decoder.layers:
            for lname, lin in find_linears(layer).items():
                lin.weight.data = quant.quantize_mx_outlier_v1(lin.weight.data, 8, 8, fi, fo, "max", 2, ax, bs, "nearest", False, False)
        d[f"{cname}|ppl"] = np.float64(ppl_of(mq, ids, SEQLEN))
        tot = 0.0
        for layer in mq.model.decoder.layers:
            for lname, lin in find_linears(layer).items():
                tot += float(lin.weight.data.double().abs().sum())
        d[f"{cname}|abs_sum"] = np.float64(tot)
        for lname, lin in find_linears(mq.model.decoder.layers[0]).items():
            d[f"{cname}|rows|{lname}"] = lin.weight.data[:ROWS].numpy().copy()
            d[f"{cname}|cols|{lname}"] = lin.weight.data[:, :ROWS].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "opt125m_fp6.npz"), **d)
    print({k: float(v) for k, v in d.items() if k.endswith("ppl") or k.endswith("ppl_fp32") or k.endswith("abs_sum")})
    import shutil
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
