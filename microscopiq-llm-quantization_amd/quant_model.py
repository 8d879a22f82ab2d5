"""utils/quant_model.py:15-72 quantize_model -- recursively replace every nn.Linear of a pretrained model by an
MXLinear carrying the same weights (the "direct" harness path: llm/opt_direct.py:98-112).  `base_model` and `lm_head`
attributes are left alone exactly like the reference (:68); nn.Conv2d has no counterpart on this hot path."""
import copy

import torch.nn as nn

from .linear import MXLinear


def quantize_model(model, mx_specs):
    if type(model) == nn.Conv2d:
        raise NotImplementedError("mx.Conv2d is outside the hot path (SURVEY.md 2 #12)")
    if type(model) == nn.Linear:
        quant_mod = MXLinear(model.in_features, model.out_features, True, mx_specs)      # :33-41 (always built with a bias)
        quant_mod = quant_mod.to(model.weight.device)
        quant_mod.weight.data = model.weight.data.clone()
        if model.bias is not None:
            quant_mod.bias.data = model.bias.data.clone()
        return quant_mod
    if type(model) in (nn.Sequential, nn.ModuleList) or isinstance(model, nn.Sequential):   # :43-59
        return nn.Sequential(*[quantize_model(m, mx_specs) for _, m in model.named_children()])
    q_model = copy.deepcopy(model)                                                        # :61-70
    for attr in dir(model):
        mod = getattr(model, attr)
        if isinstance(mod, nn.Module) and attr != 'base_model' and attr != 'lm_head':
            setattr(q_model, attr, quantize_model(mod, mx_specs))
    return q_model
