"""utils/modelutils.py:5-15"""
import torch
import torch.nn as nn

DEV = torch.device('cuda:0')


def find_layers(module, layers=[nn.Conv2d, nn.Linear], name=''):
    """{qualified name: module} of every Linear / Conv2d below `module` (utils/modelutils.py:8-15)."""
    if type(module) in layers:
        return {name: module}
    found = {}
    for child_name, child in module.named_children():
        found.update(find_layers(child, layers=layers, name=name + '.' + child_name if name != '' else child_name))
    return found
