"""Model helpers of the harness (the role of utils/modelutils.py:5-15)."""
import torch
import torch.nn as nn

DEV = torch.device('cuda:0')


def find_layers(module, layers=(nn.Conv2d, nn.Linear), name=''):
    """{qualified name: layer} for every module below `module` (itself included, under `name`) whose exact type is
    listed in `layers`, in registration order; the search does not descend into a matching module -- the
    contract of utils/modelutils.py:8-15, here as one walk over `named_modules` instead of a recursion."""
    wanted = tuple(layers)
    hits, closed = {}, []
    for path, sub in module.named_modules():
        if any(c == '' or path == c or path.startswith(c + '.') for c in closed):
            continue                                  # inside a layer that was already taken
        if type(sub) in wanted:
            hits[(name + '.' + path if name and path else name or path)] = sub
            closed.append(path)
    return hits
