"""llm/llama.py surface: get_llama (:20-58) and llama_eval (:176-284)."""
import types

import torch

from .evalppl import perplexity, quantize_layers_nearest


def get_llama(model, model_type="llama"):
    """llm/llama.py:20-58: load without re-initialising weights; seqlen fixed to 2048 (:57).  The reference swaps
    torch.nn.init's initialisers for no-ops and leaves them swapped for the rest of the process; here they are put back once the
    checkpoint is loaded (every nn.Linear built afterwards would otherwise hold uninitialised memory)."""
    def skip(*_, **__):
        pass
    saved = (torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_)
    torch.nn.init.kaiming_uniform_ = skip
    torch.nn.init.uniform_ = skip
    torch.nn.init.normal_ = skip
    try:
        if model_type == "mistral":
            from transformers import MistralForCausalLM as cls
        elif model_type == "mixtral":
            from transformers import MixtralForCausalLM as cls
        else:
            from transformers import LlamaForCausalLM as cls
        m = cls.from_pretrained(model, torch_dtype="auto")
    finally:
        torch.nn.init.kaiming_uniform_, torch.nn.init.uniform_, torch.nn.init.normal_ = saved
    m.seqlen = 2048
    return m


@torch.no_grad()
def llama_eval(model, testenc, dev, args=None, quant_cfg=None):
    """llm/llama.py:176-284.  `args.nearest` selects the RTN path (:226-253); returns the perplexity
    (the reference prints it, :282)."""
    print('Evaluating ...')
    args = args or types.SimpleNamespace(nearest=True, use_mx=True)
    use_cache = model.config.use_cache
    model.config.use_cache = False
    if getattr(args, "nearest", False):
        if not getattr(args, "use_mx", True):
            raise TypeError("Quantizer.configure() got MX keyword arguments (llm/llama.py:229-237 needs --use-mx)")
        quantize_layers_nearest(model.model.layers, dev, quant_cfg, log=print)
    ppl = perplexity(model, testenc, dev, model.seqlen)
    print(ppl)
    model.config.use_cache = use_cache
    return ppl
