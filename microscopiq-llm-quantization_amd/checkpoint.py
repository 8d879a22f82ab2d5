"""On-disk format for packed models -- what the reference's vestigial ``--save`` / ``--load`` would have
written (llm/opt.py:510-512 ``torch.save(model.state_dict(), args.save)`` after ``opt_pack3``; :287-294
``load_quant3``: build the model, ``make_quant3`` every Linear, ``load_state_dict``).

One ``.safetensors`` file: the packed planes of every QuantLinear (uint8) plus every other parameter /
buffer of the model, and a JSON header in the metadata describing each packed layer, so that a model can be
rebuilt without re-quantising (and, for the 70B row-parallel configuration, shard by shard: every rank saves
and loads its own file, `shard` / `world_size` are recorded).

    header = {"format": "msq-packed", "version": 2, "shard": r, "world_size": G,
              "layers": {"<module name>": {"in_features", "out_features", "block_size", "layout",
                                            "in_kind", "out_kind", "inlier_elem_format",
                                            "outlier_elem_format", "bias", "out_dtype"}}}

Version 2 (round 3 onwards) added: "fused" layer entries (q / k / v and gate / up packed as one weight, state_dict keys under
``<first sibling>.fused.proj.*``), planes of off-grid shapes padded to the tile grid ("padded_out" / "padded_in" record the
padded shape) and "values" layers for modules built from an existing PackedWeight.  A version-1 file has none of these and
still loads; a file with a HIGHER version than this build's is refused by read_header with a clear message instead of a
misleading "layer does not exist" / size mismatch.

MX-operand layers (MXLinearW4A8: codes in the operand order of the scaled MFMA) are listed with
``"layout": "mx-operand"`` and ``"w_fmt"`` ("e2m1" plain MX-FP4, "e4m3" exactly packed fake-quant values).
"""
import json

import torch
import torch.nn as nn

from ._lib import MsqError
from .qlinear import K_MULT, N_MULT, FusedProjections, MXLinearW4A8, ProjectionSlice, QuantLinear, _ceil_to

FORMAT, VERSION = "msq-packed", 2


def _packed_layers(model):
    return {name: m for name, m in model.named_modules() if isinstance(m, (QuantLinear, MXLinearW4A8))}


def _fused_groups(model):
    """packed module name ("<parent>.<first>.fused.proj") -> (parent name, [sibling names in slice order], splits)"""
    slices = {}
    for name, m in model.named_modules():
        if isinstance(m, ProjectionSlice):
            slices.setdefault(id(m.shared()), []).append((m.index, name))
    groups = {}
    for name, m in model.named_modules():
        if isinstance(m, FusedProjections) and id(m) in slices:
            sibs = [n for _, n in sorted(slices[id(m)])]
            parent = sibs[0].rsplit(".", 1)[0] if "." in sibs[0] else ""
            groups[name + ".proj"] = (parent, [n.rsplit(".", 1)[-1] for n in sibs], list(m.splits))
    return groups


def save_packed(model, path, shard=0, world_size=1):
    """Write every tensor of `model` (packed planes included) and the layer table to `path`."""
    from safetensors.torch import save_file
    layers = {}
    fused = _fused_groups(model)
    for name, m in _packed_layers(model).items():
        if isinstance(m, MXLinearW4A8):
            layers[name] = dict(in_features=m.in_features, out_features=m.out_features, layout="mx-operand", w_fmt=m.w_fmt,
                                bias=m.bias is not None, out_dtype=str(m.out_dtype).replace("torch.", ""))
            continue
        layers[name] = dict(in_features=m.in_features, out_features=m.out_features, block_size=m.block_size,
                            layout=m.layout, in_kind=m.in_kind, out_kind=m.out_kind,
                            inlier_elem_format=m.inlier_elem_format, outlier_elem_format=m.outlier_elem_format,
                            bias=m.bias is not None, out_dtype=str(m.out_dtype).replace("torch.", ""),
                            padded_out=_ceil_to(m.out_features, N_MULT), padded_in=_ceil_to(m.in_features, K_MULT))
    for name, (parent, sibs, splits) in fused.items():            # fused q / k / v or gate / up: one packed weight, several stand-ins
        layers[name]["fused"] = dict(parent=parent, names=sibs, splits=splits)
    header = dict(format=FORMAT, version=VERSION, shard=int(shard), world_size=int(world_size), layers=layers)
    tensors = {k: v.detach().contiguous().cpu() for k, v in model.state_dict().items()}
    save_file(tensors, path, metadata={"msq": json.dumps(header)})
    return header


def read_header(path):
    from safetensors import safe_open
    with safe_open(path, framework="pt") as f:
        meta = f.metadata() or {}
    if "msq" not in meta:
        raise MsqError("%s is not an msq-packed checkpoint (no header)" % path)
    header = json.loads(meta["msq"])
    if header.get("format") != FORMAT or int(header.get("version", -1)) > VERSION:
        raise MsqError("unsupported checkpoint format %r version %r (this build reads %s versions 1..%d)"
                       % (header.get("format"), header.get("version"), FORMAT, VERSION))
    return header


def _swap(model, name, new):
    parent = model
    parts = name.split(".")
    for p in parts[:-1]:
        parent = getattr(parent, p)
    setattr(parent, parts[-1], new)


def load_packed(model, path, device=None, strict=True):
    """`model` is the freshly constructed (unquantised) architecture: every layer listed in the header is
    replaced by an empty QuantLinear of the recorded configuration (llm/opt.py:287 ``make_quant3``), then
    all tensors are loaded.  Returns the header."""
    from safetensors.torch import load_file
    header = read_header(path)
    modules = dict(model.named_modules())

    def empty_module(name, d, dev):
        if d.get("layout") == "mx-operand":
            return MXLinearW4A8(d["in_features"], d["out_features"], d["bias"], getattr(torch, d["out_dtype"]), dev, w_fmt=d["w_fmt"])
        if d["inlier_elem_format"] == "values":          # packed from dense values (QuantLinear.from_dense)
            q = QuantLinear.empty_single_plane(d["in_features"], d["out_features"], d["bias"], d["out_kind"],
                                               getattr(torch, d["out_dtype"]), dev)
        else:
            q = QuantLinear(d["in_features"], d["out_features"], d["bias"], d["block_size"], d["inlier_elem_format"],
                            d["outlier_elem_format"], getattr(torch, d["out_dtype"]), device=dev, layout=d["layout"])
        if (q.in_kind, q.out_kind) != (d["in_kind"], d["out_kind"]):
            raise MsqError("checkpoint layer %r was packed with plane kinds %r, this build derives %r" % (
                name, (d["in_kind"], d["out_kind"]), (q.in_kind, q.out_kind)))
        return q

    for name, d in header["layers"].items():
        if "fused" in d:                                  # one packed weight behind several sibling Linears
            f = d["fused"]
            parent = modules.get(f["parent"]) if f["parent"] else model
            olds = [getattr(parent, n, None) for n in f["names"]] if parent is not None else [None]
            if parent is None or any(o is None for o in olds):
                raise MsqError("checkpoint layer %r: the model has no %r under %r" % (name, f["names"], f["parent"]))
            for o, n_out in zip(olds, f["splits"]):
                if (getattr(o, "in_features", None), getattr(o, "out_features", None)) != (d["in_features"], n_out):
                    raise MsqError("checkpoint layer %r does not match the model's %r" % (name, f["names"]))
            o0 = olds[0]
            dev = device if device is not None else next(o0.parameters(), next(o0.buffers(), torch.zeros(0))).device
            fused = FusedProjections(empty_module(name, d, dev), f["splits"])
            for i, n in enumerate(f["names"]):
                setattr(parent, n, ProjectionSlice(fused, i, owner=(i == 0)))
            continue
        old = modules.get(name)
        if old is None:
            raise MsqError("checkpoint layer %r does not exist in the model" % name)
        if isinstance(old, (nn.Linear, QuantLinear, MXLinearW4A8)):
            if (old.in_features, old.out_features) != (d["in_features"], d["out_features"]):
                raise MsqError("checkpoint layer %r has shape %dx%d, the model %dx%d" % (
                    name, d["out_features"], d["in_features"], old.out_features, old.in_features))
        dev = device if device is not None else next(old.parameters(), next(old.buffers(), torch.zeros(0))).device
        _swap(model, name, empty_module(name, d, dev))
    state = load_file(path)
    model.load_state_dict(state, strict=strict)
    return header
