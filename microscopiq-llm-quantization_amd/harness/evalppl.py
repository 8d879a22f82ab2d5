"""Shared pieces of llama_eval / opt_eval: in-place RTN quantisation of every Linear inside the
decoder layers (embeddings and lm_head untouched, llm/llama.py:262-272) and the reference's
perplexity formula (llm/llama.py:264-282):

    nsamples = numel // seqlen                      (tail dropped)
    nll_i    = CrossEntropy(logits_i[:-1], tok_i[1:]) * seqlen
    ppl      = exp( sum_i nll_i / (nsamples * seqlen) )

The reference evaluates layer by layer with CPU offload (llm/llama.py:195-260); that loop calls
decoder layers with the transformers-4.26 signature and does not run on current transformers, so
the windows are pushed through the whole model instead -- the same numbers, no offload."""
import torch
import torch.nn as nn

from ..quant import MXQuantizer, quantize_mx_outlier_v1
from .modelutils import find_layers

# llm/llama.py:229-237, llm/opt.py:193-201: the harness hard-codes this configuration
DEFAULT_QUANT = dict(inlier_scale_bits=8, outlier_scale_bits=8, inlier_elem_format='int2',
                     outlier_elem_format='fp4', axes=[0], block_size=16)


@torch.no_grad()
def quantize_layers_nearest(layers, dev, quant_cfg=None, log=None):
    """RTN path (`--nearest`, llm/llama.py:226-253): weight.data <- quantize_mx_outlier_v1(weight)."""
    cfg = dict(DEFAULT_QUANT)
    if quant_cfg:
        cfg.update(quant_cfg)
    for i, layer in enumerate(layers):
        layer.to(dev)
        subset = find_layers(layer)
        for name in subset:
            quantizer = MXQuantizer()
            quantizer.configure(**cfg)
            W = subset[name].weight.data
            quantizer.find_params(W, weight=True)
            subset[name].weight.data = quantize_mx_outlier_v1(
                W, quantizer.inlier_scale_bits, quantizer.outlier_scale_bits, quantizer.inlier_elem_format,
                quantizer.outlier_elem_format, quantizer.shared_exp_method, quantizer.std_dev, quantizer.axes,
                quantizer.block_size, quantizer.round, quantizer.flush_fp32_subnorms, quantizer.custom_cuda
            ).to(next(iter(layer.parameters())).dtype)
        if log:
            log(i)


LLAMA_FUSE = [("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj")]      # siblings that read the same input


@torch.no_grad()
def pack_layers(layers, path="bf16", fuse=None):
    """After quantize_layers_nearest / the GPTQ pass: swap every (already fake-quantised) Linear inside the decoder
    layers for a packed QuantLinear (what ``opt_pack3`` / ``make_quant3`` were meant to do, llm/opt.py:255-264).
    The weights are stored in the smallest exact single-plane kind (8.25 bits/weight for the harness default
    int2 / fp4 configuration) and the forward runs the fused dequant-GEMM.  EVERY nn.Linear that find_layers returns is
    packed (utils/modelutils.py:8-15): shapes off the kernels' tile grid are zero-padded at pack time (qlinear._pad2d).
    Returns (packed, kept_dense); kept_dense counts only Linears that are not on the GPU.

    ``fuse`` (e.g. LLAMA_FUSE): groups of sibling Linears that read the same input are packed as one weight
    (qlinear.fuse_projections): one GEMM launch per group, the same values.

    ``path="mx"``: W4A8 on the MX matrix path instead -- the fake-quant values become one exact e4m3 operand
    (MXLinearW4A8.from_values, 8.25 bits/weight), the forward quantises the activations to MX-FP8 (plain OCP-MX,
    block 32) and multiplies on the scaled MFMA.  A layer whose values do not fit e4m3 x 2^s per 32-block (e.g. posit
    outliers) stays on the bf16-activation QuantLinear."""
    from .._lib import MsqError
    from ..qlinear import MXLinearW4A8, fuse_projections, make_quant
    packed = dense = 0

    def set_child(root, name, m):
        parent = root
        parts = name.split(".")
        for p_ in parts[:-1]:
            parent = getattr(parent, p_)
        setattr(parent, parts[-1], m)

    for layer in layers:
        names = {}
        for name, lin in find_layers(layer, layers=[nn.Linear]).items():
            if not lin.weight.is_cuda:
                dense += 1
                continue
            names[name] = None
            packed += 1
        if path == "mx":
            for mod_name, mod in list(layer.named_modules()):
                kids = dict(mod.named_children())
                for group in (fuse or ()):
                    fulls = [(mod_name + "." + g if mod_name else g) for g in group]
                    if all(g in kids and isinstance(kids[g], nn.Linear) for g in group) and all(f in names for f in fulls):
                        try:
                            fuse_projections(mod, list(group), None, path="mx")
                            for f in fulls:
                                del names[f]
                        except MsqError:
                            pass
            for name in list(names):
                lin = dict(layer.named_modules())[name]
                try:
                    m = MXLinearW4A8.from_values(lin.weight.data, lin.bias, out_dtype=lin.weight.dtype if lin.weight.dtype == torch.bfloat16 else torch.float32)
                except MsqError:
                    continue
                set_child(layer, name, m)
                del names[name]
        make_quant(layer, names, fuse=fuse)
    return packed, dense


@torch.no_grad()
def perplexity(model, testenc, dev, seqlen, fp32_loss=False):
    """llm/llama.py:264-282.  On an fp16 model the reference's cross entropy is an fp16 number (spacing 2^-7 around ln 72000 = 11.2:
    the perplexity then moves in steps of 0.8 %); ``fp32_loss`` upcasts the logits first -- NOT the reference's arithmetic, for
    comparisons finer than that step (bench.py llama7b_e2e)."""
    ids = testenc.input_ids if hasattr(testenc, "input_ids") else testenc
    nsamples = ids.numel() // seqlen
    model.to(dev)
    nlls = []
    loss_fct = nn.CrossEntropyLoss()
    for i in range(nsamples):
        batch = ids[:, (i * seqlen):((i + 1) * seqlen)].to(dev)
        lm_logits = model(batch).logits
        if fp32_loss:
            lm_logits = lm_logits.float()
        shift_logits = lm_logits[:, :-1, :].contiguous()
        shift_labels = batch[:, 1:]
        loss = loss_fct(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
        nlls.append(loss.float() * seqlen)
    return torch.exp(torch.stack(nlls).sum() / (nsamples * seqlen)).item()
