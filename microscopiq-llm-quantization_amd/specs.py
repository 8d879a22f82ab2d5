"""MX configuration dictionary -- the surface of number_system/mx/specs.py
(MxSpecs :61-179, get_default_mx_specs, apply_mx_specs, finalize_mx_specs :276-319,
mx_assert_test :343-355) that the Linear path reads.  Host-side logic only."""
import collections
import os
import traceback

_ASSERT_MODE = os.environ.get('MX_ASSERT', 'False')        # specs.py:58

_DEFAULTS = collections.OrderedDict([
    ("scale_bits", 0),
    ("w_elem_format", None), ("a_elem_format", None),
    ("w_elem_format_bp", None), ("a_elem_format_bp_ex", None), ("a_elem_format_bp_os", None),
    ("mx_flush_fp32_subnorms", False),
    ("shared_exp_method", "max"), ("block_size", 0),
    ("bfloat", 0), ("fp", 0), ("bfloat_subnorms", True),
    ("quantize_backprop", True),
    ("round", "nearest"), ("round_m", "nearest"), ("round_weight", "nearest"), ("round_output", "nearest"),
    ("round_grad_weight", "nearest"), ("round_grad_input", "nearest"), ("round_mx_output", "nearest"),
    ("round_mx_input_grad_input", "nearest"), ("round_mx_weight_grad_input", "nearest"),
    ("round_mx_grad_output_grad_input", "nearest"), ("round_mx_input_grad_weight", "nearest"),
    ("round_mx_grad_output_grad_weight", "nearest"),
    ("softmax_exp2", False), ("vec_use_exp2", False), ("vec_use_recip", False),
    ("custom_cuda", False),
])

_DERIVED = [
    ("w_elem_format_bp", "w_elem_format"), ("a_elem_format_bp_os", "a_elem_format"),
    ("a_elem_format_bp_ex", "a_elem_format"),
    ("round_m", "round"), ("round_output", "round"), ("round_grad_weight", "round"),
    ("round_grad_input", "round"), ("round_weight", "round"), ("round_mx_output", "round"),
    ("round_mx_input_grad_input", "round_grad_input"), ("round_mx_weight_grad_input", "round_grad_input"),
    ("round_mx_grad_output_grad_input", "round_grad_input"), ("round_mx_input_grad_weight", "round_grad_input"),
    ("round_mx_grad_output_grad_weight", "round_grad_input"),
]


class MxSpecs(collections.UserDict):
    """Dictionary of quantisation parameters with the reference's keys and defaults."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in _DEFAULTS.items():
            self.data.setdefault(k, v)
        for k in self.data:
            assert k in _DEFAULTS, "unknown mx spec %r" % k


def get_default_mx_specs():
    return MxSpecs()


def apply_mx_specs(mx_specs, default_mx_specs=None):
    """Overlay user specs on the defaults; None stays None (= no quantisation)."""
    if mx_specs is None:
        return None
    out = default_mx_specs if default_mx_specs is not None else get_default_mx_specs()
    for k, v in dict(mx_specs).items():
        if k not in out:
            raise KeyError("Unknown key %r passed to mx specs" % k)
        if v is not None or out[k] is None:
            out[k] = v
    return out


def get_backwards_mx_specs(specs):
    bspecs = MxSpecs(dict(specs))
    if bspecs["quantize_backprop"] is False:
        for k in ("w_elem_format_bp", "a_elem_format_bp_ex", "a_elem_format_bp_os"):
            bspecs[k] = None
        bspecs["bfloat"] = 0
        bspecs["fp"] = 0
    return bspecs


def finalize_mx_specs(specs, early_exit=True):
    """specs.py:276-319: resolve dependent specs; None when nothing is quantised."""
    quantised = any(specs.get(k, 0) for k in ("w_elem_format", "a_elem_format", "w_elem_format_bp",
                                              "a_elem_format_bp_os", "a_elem_format_bp_ex", "bfloat", "fp"))
    if not quantised and early_exit:
        return None
    for dst, src in _DERIVED:
        if (dst not in specs or specs[dst] is None) and src in specs:
            specs[dst] = specs[src]
    return apply_mx_specs(specs, get_default_mx_specs())


def mx_assert_test(mx_specs):
    if _ASSERT_MODE == "True" and mx_specs is None:
        stack = traceback.extract_stack()
        raise ValueError("mx_specs is None under MX_ASSERT (called from %s)" % stack[-3].name)
