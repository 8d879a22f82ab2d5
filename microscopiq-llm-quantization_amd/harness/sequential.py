"""Layer-sequential GPTQ calibration -- llm/llama.py:62-173 llama_sequential and llm/opt.py:26-128 opt_sequential.

Same flow as the reference: capture what reaches decoder layer 0 for every calibration sample; then, layer by layer,
(a) hook every Linear of the current subset and run the layer on its inputs so that each ``GPTQ.add_batch`` accumulates
its Hessian (:127-135), (b) ``fasterquant`` every Linear of the subset (:137-145), (c) re-run the quantised layer to
produce the inputs of the next one (:147-157).  With ``true_sequential`` the attention / MLP projections are calibrated
in the four groups of :116-121, later groups seeing the quantised earlier ones.

The reference calls the decoder layer with the transformers-4.26 keyword set (attention_mask, position_ids) and does
not run on current transformers; here the positional and keyword arguments layer 0 really receives are recorded by a
forward pre-hook and replayed, so the driver does not depend on the layer signature.  Layers are moved to `dev` one at
a time like the reference (CPU offload of everything else)."""
import types

import torch
import torch.nn as nn

from ..quant import MXQuantizer, Quantizer
from .evalppl import DEFAULT_QUANT
from .gptq import GPTQ
from .modelutils import find_layers

LLAMA_TRUE_SEQUENTIAL = [['self_attn.k_proj', 'self_attn.v_proj', 'self_attn.q_proj'], ['self_attn.o_proj'],
                         ['mlp.up_proj', 'mlp.gate_proj'], ['mlp.down_proj']]                     # llm/llama.py:116-121


class _Stop(Exception):
    pass


def _default_args(args):
    a = types.SimpleNamespace(nsamples=128, true_sequential=False, use_mx=True, percdamp=.01, groupsize=-1, act_order=False,
                              static_groups=False)
    for k, v in (vars(args) if args is not None else {}).items():
        setattr(a, k, v)
    return a


def _first(out):
    return out[0] if isinstance(out, (tuple, list)) else out


@torch.no_grad()
def _sequential(model, layers, prefix, pre_modules, dataloader, dev, args, true_sequential_groups, quant_cfg, log):
    args = _default_args(args)
    use_cache = model.config.use_cache
    model.config.use_cache = False
    for m in pre_modules:
        m.to(dev)
    layers[0].to(dev)
    captured = []

    def catch(module, a, kw):
        captured.append((tuple(t.detach() if torch.is_tensor(t) else t for t in a),
                         {k: (v.detach() if torch.is_tensor(v) else v) for k, v in kw.items()}))
        raise _Stop()

    h = layers[0].register_forward_pre_hook(catch, with_kwargs=True)
    n = 0
    for batch in dataloader:
        if n >= args.nsamples:
            break
        try:
            model(batch[0].to(dev))
        except _Stop:
            pass
        n += 1
    h.remove()
    layers[0].cpu()
    for m in pre_modules:
        m.cpu()
    torch.cuda.empty_cache()
    inps = [c[0][0] for c in captured]                       # hidden states [1, seq, hidden] per sample
    rest = [(c[0][1:], c[1]) for c in captured]              # whatever else the layer is called with, replayed as is
    if log:
        log('Ready.')
    cfg = dict(DEFAULT_QUANT)                                # llm/llama.py:102-113: hard-coded in the reference
    if quant_cfg:
        cfg.update(quant_cfg)
    quantizers = {}
    for i in range(len(layers)):
        layer = layers[i].to(dev)
        full = find_layers(layer)
        groups = [[nm for nm in g if nm in full] for g in true_sequential_groups] if (args.true_sequential and true_sequential_groups) else [list(full.keys())]
        run = lambda j: _first(layer(inps[j], *rest[j][0], **rest[j][1]))
        for names in groups:
            subset = {nm: full[nm] for nm in names}
            gptq = {}
            for name in subset:
                gptq[name] = GPTQ(subset[name])
                gptq[name].quantizer = MXQuantizer() if args.use_mx else Quantizer()
                gptq[name].quantizer.configure(**cfg)

            def add_batch(name):
                def tmp(_, inp, out):
                    gptq[name].add_batch(inp[0].data, out.data)
                return tmp

            handles = [subset[name].register_forward_hook(add_batch(name)) for name in subset]
            for j in range(len(inps)):
                run(j)
            for hd in handles:
                hd.remove()
            for name in subset:
                if log:
                    log("%d %s" % (i, name)); log('Quantizing ...')
                gptq[name].fasterquant(percdamp=args.percdamp, groupsize=args.groupsize, actorder=args.act_order,
                                       static_groups=args.static_groups, verbose=bool(log))
                quantizers['%s.%d.%s' % (prefix, i, name)] = gptq[name].quantizer
                gptq[name].free()
        inps = [run(j) for j in range(len(inps))]            # outputs of the quantised layer feed the next one
        layers[i] = layer.cpu()
        del layer
        torch.cuda.empty_cache()
    model.config.use_cache = use_cache
    return quantizers


def llama_sequential(model, dataloader, dev, args=None, quant_cfg=None, log=print):
    """llm/llama.py:62-173; `args`: nsamples, true_sequential, use_mx, percdamp, groupsize, act_order, static_groups."""
    if log:
        log('Starting ...')
    m = model.model
    pre = [m.embed_tokens, m.norm] + ([m.rotary_emb] if hasattr(m, "rotary_emb") else [])
    return _sequential(model, m.layers, 'model.layers', pre, dataloader, dev, args, LLAMA_TRUE_SEQUENTIAL, quant_cfg, log)


def opt_sequential(model, dataloader, dev, args=None, quant_cfg=None, log=print):
    """llm/opt.py:26-128"""
    if log:
        log('Starting ...')
    d = model.model.decoder
    pre = [d.embed_tokens, d.embed_positions]
    for nm in ('project_out', 'project_in', 'final_layer_norm'):
        if getattr(d, nm, None) is not None:
            pre.append(getattr(d, nm))
    return _sequential(model, d.layers, 'model.decoder.layers', pre, dataloader, dev, args, None, quant_cfg, log)
