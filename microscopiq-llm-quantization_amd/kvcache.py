"""KV-cache quantisation at the GEAR hook (BASELINE config 4).

Reference surface (kv_quant/GEARLM/Simulated): the two group fake-quant functions of compress_function.py:8-70, the
dispatcher ``compress_insert_function`` (:428-517) for the methods that only use them (KIVI, kcvtQfixed, tokenQfixed,
channelQfixed, Flexgen), ``CompressionConfig`` (compress_config.py) and the streaming logic of the attention hook
(modeling_llama_new.py:944-1030) -- all executed by libmsq_hip.so on the cache tensor in its own [B, H, S, D] layout.
The reference has no MX code in kv_quant/ at all; the MX variants here are new:

  method "MX"   plain OCP-MX fake-quant (mx_ops.py:332-457): K with blocks along the tokens of a channel, V with blocks
                along head_dim of a token -- the axes KIVI uses for its integer groups;
  method "MSQ"  the MicroScopiQ outlier-aware quantiser (utils/quant.py:147-266) on the same axes.

``MXKVCache`` is a transformers Cache whose layers re-compress every ``streaming_gap`` tokens exactly like the
reference's hook: at prefill the first ``L - L % gap`` prompt tokens are compressed before they are used, during decoding
the last ``gap`` tokens are compressed whenever the cached length reaches a multiple of ``gap``.
The low-rank / sparse GEAR methods (gearl*, gearsl*) are third-party algorithms outside the hot path and raise."""
import torch

from ._lib import MsqError, check, current_stream, lib, ptr
from .mx_ops import _quantize_mx
from .quant import outlier_fakequant

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _group_quant(input, quantize_bit, group_size, along_tokens):
    if not input.is_cuda:
        raise MsqError("KV-cache quantisation needs a CUDA/HIP tensor (no CPU fallback)")
    if input.ndim != 4:
        raise MsqError("expected a [batch, heads, seq, head_dim] cache tensor")
    if input.dtype not in _DT:
        raise MsqError("cache dtype must be float32, float16 or bfloat16")
    B, H, S, D = input.shape
    if along_tokens:
        if group_size <= 0 or S % group_size:
            # the reference's .view(batch, seq // group, group, H * D) raises for a ragged sequence (:50-53)
            raise RuntimeError("shape '[%d, %d, %d, %d]' is invalid for input of size %d"
                               % (B, S // max(group_size, 1), group_size, H * D, input.numel()))
    elif group_size <= 0 or (H * D) % group_size:
        raise ValueError("group_size should be a factor of the last dimension size")          # :16-17
    x = input.contiguous()
    out = torch.empty_like(x)
    check(lib().msq_kv_group_quant(ptr(x), ptr(out), _DT[x.dtype], B, H, S, D, int(quantize_bit), int(group_size),
                                   int(bool(along_tokens)), current_stream(x.device)), "msq_kv_group_quant")
    return out


def fake_groupwise_token_asymmetric_quantization(input, quantize_bit, group_size=128):
    """compress_function.py:8-38 -- asymmetric min / max groups along head.dim of every token."""
    return _group_quant(input, quantize_bit, group_size, False)


def fake_groupwise_channel_asymmetric_quantization_new(input, quantize_bit, group_size=128):
    """compress_function.py:41-70 -- asymmetric min / max groups along the tokens of every channel."""
    return _group_quant(input, quantize_bit, group_size, True)


def _msq_f32(t, scale_bits, elem_format, outlier_format, std_dev, axis, block_size):
    """The MicroScopiQ quantiser on a cache tensor, computed in FLOAT32 and cast back to the cache dtype.  Not in the cache's own half
    precision: utils/quant.py:489-492 tests the SIGNED values against bounds derived from |A|, so a block whose values are all negative -- a
    key channel with a consistent sign, common in a KV cache -- is flagged outlier as a whole; its inlier part is all zero, e_in clamps to -20
    (:207-211), the outliers enter their own domain as o 2^-20, and e_out = floor(log2(max)) - 8 lands near -25: 2^-25 is zero in fp16, the
    division gives Inf / NaN and the reference's NaN assert (:225-250) fires.  Found by running config 4's loop end to end (round 6: every
    generation with the fp16 cache came back empty).  In float32 the same block quantises to its fp8 outlier values."""
    # (fp16 / bf16 caches are read and written as they are by the float32 kernels: no cast passes, the bits of upcast -> quantise -> downcast)
    return outlier_fakequant(t, scale_bits, scale_bits, elem_format, outlier_format, std_dev, axis, block_size, compute_dtype="float32")["out"]


def mx_quantize_keys(key, elem_format="fp8_e4m3", block_size=32, scale_bits=8, outlier_format=None, std_dev=2):
    """K cache [B, H, S, D]: MX blocks of `block_size` consecutive TOKENS of one channel (per-channel grouping, the
    axis KIVI quantises keys along).  ``outlier_format`` set -> MicroScopiQ inlier / outlier quantiser instead of plain
    MX.  S need not be a multiple of the block: the last block is zero padded like every MX tensor (utils/quant.py:563-583)."""
    if outlier_format is None:
        return _quantize_mx(key, scale_bits, elem_format, axes=[2], block_size=block_size)
    return _msq_f32(key, scale_bits, elem_format, outlier_format, std_dev, 2, block_size)


def mx_quantize_values(value, elem_format="fp8_e4m3", block_size=32, scale_bits=8, outlier_format=None, std_dev=2):
    """V cache [B, H, S, D]: MX blocks along head_dim of one token (per-token grouping)."""
    if outlier_format is None:
        return _quantize_mx(value, scale_bits, elem_format, axes=[3], block_size=block_size)
    return _msq_f32(value, scale_bits, elem_format, outlier_format, std_dev, 3, block_size)


class CompressionConfig(dict):
    """compress_config.py:1-99: one entry per attention layer after copy_for_all_attention().  Only the fields the hot
    path reads are kept; ``mx_format`` / ``mx_outlier_format`` / ``mx_block`` configure the MX methods."""

    _PER_LAYER = ("compress_method", "quantize_bit", "group_size", "start_saving", "locality_saving", "token_preserving",
                  "streaming", "streaming_gap", "stream_grouping", "mx_format", "mx_outlier_format", "mx_block")

    def __init__(self, compress_method=None, attention_number=12, quantize_bit=0, group_size=0, start_saving=0,
                 locality_saving=0, token_preserving=False, streaming=False, streaming_gap=0, stream_grouping=False,
                 mx_format="fp8_e4m3", mx_outlier_format=None, mx_block=32, **ignored):
        super().__init__()
        self.compress_method, self.attention_number = compress_method, attention_number
        self.quantize_bit, self.group_size = quantize_bit, group_size
        self.start_saving, self.locality_saving, self.token_preserving = start_saving, locality_saving, token_preserving
        self.streaming, self.streaming_gap, self.stream_grouping = streaming, streaming_gap, stream_grouping
        self.mx_format, self.mx_outlier_format, self.mx_block = mx_format, mx_outlier_format, mx_block

    def create_attention_config(self, config):
        return [config for _ in range(self.attention_number)]

    def copy_for_all_attention(self):
        for f in self._PER_LAYER:
            v = getattr(self, f)
            if not isinstance(v, list):
                setattr(self, f, self.create_attention_config(v))
        return self


_INT_METHODS = ("channelQfixed", "tokenQfixed", "kcvtQfixed", "KIVI", "Flexgen")


def compress_insert_function(previous_key, previous_value, compress_config, layer_idx, pbase1=None, qbase1=None,
                             pbase2=None, qbase2=None, prefill=None):
    """compress_function.py:428-517 for the group-quant methods (+ "MX" / "MSQ"): quantises a token range of the key /
    value tensors IN PLACE and returns them."""
    cfg = compress_config
    batch, num_head, seq_len, sep_dim = previous_key.shape
    method = cfg.compress_method[layer_idx]
    if cfg.token_preserving[layer_idx] == True:                                          # noqa: E712 (:440-445)
        starting_idx = int(cfg.start_saving[layer_idx] * seq_len)
        locality_idx = int(cfg.locality_saving[layer_idx] * seq_len)
    else:
        starting_idx, locality_idx = 0, -seq_len
    bits = cfg.quantize_bit[layer_idx]
    sl = slice(starting_idx, -locality_idx)                                              # [start : -locality]
    n = len(range(*sl.indices(seq_len)))
    tok = fake_groupwise_token_asymmetric_quantization
    chn = fake_groupwise_channel_asymmetric_quantization_new

    def put(t, fn, *a):
        if t is not None and n > 0:
            t[:, :, sl, :] = fn(t[:, :, sl, :], *a)

    if method == "channelQfixed":                                                         # :447-458 (group = seq_len)
        put(previous_key, chn, bits, seq_len)
        put(previous_value, chn, bits, seq_len)
    elif method == "tokenQfixed":                                                         # :461-472
        put(previous_key, tok, bits, int(num_head * sep_dim))
        put(previous_value, tok, bits, int(num_head * sep_dim))
    elif method == "kcvtQfixed":                                                          # :475-486
        put(previous_key, chn, bits, seq_len)
        put(previous_value, tok, bits, int(num_head * sep_dim))
    elif method == "KIVI":                                                                # :489-502
        put(previous_key, chn, bits, cfg.group_size[layer_idx])
        put(previous_value, tok, bits, cfg.group_size[layer_idx])
    elif method == "Flexgen":                                                             # :504-516
        gs = cfg.group_size[layer_idx]
        keep = seq_len - seq_len % gs if seq_len % gs else 0          # [0 : -residual]; residual 0 is the empty slice [0:-0]
        if keep > 0:
            previous_key[:, :, 0:keep, :] = chn(previous_key[:, :, 0:keep, :], bits, gs)
            previous_value[:, :, 0:keep, :] = chn(previous_value[:, :, 0:keep, :], bits, gs)
    elif method in ("MX", "MSQ"):
        fmt, blk = cfg.mx_format[layer_idx], cfg.mx_block[layer_idx]
        ofmt = cfg.mx_outlier_format[layer_idx] if method == "MSQ" else None
        if method == "MSQ" and ofmt is None:
            raise MsqError("method MSQ needs mx_outlier_format")
        put(previous_key, mx_quantize_keys, fmt, blk, 8, ofmt)
        put(previous_value, mx_quantize_values, fmt, blk, 8, ofmt)
    elif method is None:
        pass
    else:
        raise NotImplementedError("compress method %r (GEAR low-rank / sparse variants) is outside the hot path" % (method,))
    return previous_key, previous_value


def _make_layer_class():
    from transformers.cache_utils import DynamicLayer

    class MXKVLayer(DynamicLayer):
        """One attention layer's cache with the streaming re-compression of modeling_llama_new.py:944-1030."""

        def __init__(self, compress_config=None, layer_idx=0):
            super().__init__()
            self.compress_config, self.layer_idx = compress_config, layer_idx

        def update(self, key_states, value_states, *args, **kwargs):
            cfg, li = self.compress_config, self.layer_idx
            if not self.is_initialized:
                self.lazy_initialization(key_states, value_states)
            active = cfg is not None and cfg.compress_method[li] is not None and cfg.streaming[li] is True
            if not active:
                return super().update(key_states, value_states, *args, **kwargs)
            gap = cfg.streaming_gap[li]
            cached = self.get_seq_length()
            if cached == 0:
                # first call for this layer: the hook sees past = (key_states, value_states) (:950-951)
                if key_states.shape[-2] > 1:                     # prefill (:942-943)
                    key_states, value_states = self._compress_prefill(key_states, value_states, gap)
            elif key_states.shape[-2] > 1:
                # a later multi-token call: the reference's `prefill` flag is set again and the whole past is treated
                # like a prompt (:959-971)
                self.keys, self.values = self._compress_prefill(self.keys, self.values, gap)
            elif cached % gap == 0:                              # decoding: the last `gap` tokens are complete (:972-977)
                if cfg.stream_grouping[li] == True:              # noqa: E712
                    k, v = self.keys[:, :, -gap:, :], self.values[:, :, -gap:, :]
                    k, v = compress_insert_function(k.contiguous(), v.contiguous(), cfg, li, prefill=False)
                    self.keys = torch.cat([self.keys[:, :, :-gap, :], k], dim=2)
                    self.values = torch.cat([self.values[:, :, :-gap, :], v], dim=2)
                else:
                    self.keys, self.values = self._compress_whole(self.keys, self.values)
            return super().update(key_states, value_states, *args, **kwargs)

        def _compress_whole(self, k, v):
            cfg, li = self.compress_config, self.layer_idx
            if cfg.compress_method[li] == "KIVI":                # :993-1000: only whole groups, the rest stays exact
                gs = cfg.group_size[li]
                fixed = k.shape[2] // gs * gs
                rk, rv = k[:, :, fixed:, :], v[:, :, fixed:, :]
                ck, cv = compress_insert_function(k[:, :, :fixed, :].contiguous(), v[:, :, :fixed, :].contiguous(), cfg, li)
                return torch.cat([ck, rk], dim=2), torch.cat([cv, rv], dim=2)
            return compress_insert_function(k.contiguous(), v.contiguous(), cfg, li)

        def _compress_prefill(self, k, v, gap):
            cfg, li = self.compress_config, self.layer_idx
            if cfg.stream_grouping[li] != True:                  # noqa: E712
                return self._compress_whole(k, v)
            seq_len = k.shape[2]
            residual = seq_len % gap                             # :961-971
            if seq_len - residual == 0:
                return k, v
            ck, cv = compress_insert_function(k[:, :, :seq_len - residual, :].contiguous(),
                                              v[:, :, :seq_len - residual, :].contiguous(), cfg, li, prefill=True)
            if residual == 0:
                return ck, cv
            return torch.cat([ck, k[:, :, -residual:, :]], dim=2), torch.cat([cv, v[:, :, -residual:, :]], dim=2)

    return MXKVLayer


def MXKVCache(compress_config, num_layers=None):
    """A transformers Cache (``past_key_values=MXKVCache(cfg)``) whose layers fake-quantise K / V at the GEAR hook."""
    from transformers.cache_utils import Cache
    layer_cls = _make_layer_class()
    n = num_layers if num_layers is not None else compress_config.attention_number
    return Cache(layers=[layer_cls(compress_config, i) for i in range(n)])
