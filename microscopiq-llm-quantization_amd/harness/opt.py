"""llm/opt.py surface: get_opt (:9-23) and opt_eval (:131-252)."""
import types

import torch

from .evalppl import perplexity, quantize_layers_nearest


def get_opt(model):
    """llm/opt.py:9-23; the no-op initialisers are put back after the load (see harness/llama.py get_llama)."""
    def skip(*_, **__):
        pass
    saved = (torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_)
    torch.nn.init.kaiming_uniform_ = skip
    torch.nn.init.uniform_ = skip
    torch.nn.init.normal_ = skip
    try:
        from transformers import OPTForCausalLM
        m = OPTForCausalLM.from_pretrained(model, torch_dtype='auto')
    finally:
        torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_ = saved
    m.seqlen = m.config.max_position_embeddings           # llm/opt.py:22
    return m


@torch.no_grad()
def opt_eval(model, testenc, dev, args=None, quant_cfg=None):
    """llm/opt.py:131-252 (RTN path :190-218)."""
    print('Evaluating ...')
    args = args or types.SimpleNamespace(nearest=True, use_mx=True)
    use_cache = model.config.use_cache
    model.config.use_cache = False
    if getattr(args, "nearest", False):
        quantize_layers_nearest(model.model.decoder.layers, dev, quant_cfg, log=print)
    ppl = perplexity(model, testenc, dev, model.seqlen)
    print(ppl)
    model.config.use_cache = use_cache
    return ppl
