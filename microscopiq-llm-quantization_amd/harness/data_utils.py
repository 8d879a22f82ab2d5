"""Calibration / evaluation token streams (utils/data_utils.py:36-56, :189-204).

`get_wikitext2` needs the HuggingFace dataset and tokenizer on local disk (there is no network in
the build image or on the GPU box); `synthetic_tokens` gives a seeded stand-in of the same shape
for smoke runs and parity fixtures."""
import random

import numpy as np
import torch


class _Enc:
    def __init__(self, ids):
        self.input_ids = ids


def synthetic_tokens(vocab_size, n_tokens, seed=0):
    g = torch.Generator().manual_seed(seed)
    return _Enc(torch.randint(0, vocab_size, (1, n_tokens), generator=g))


def _wikitext2_split(split):
    from datasets import load_dataset
    return "\n\n".join(load_dataset('wikitext', 'wikitext-2-raw-v1', split=split)['text'])


def get_wikitext2(nsamples, seed, seqlen, model):
    """utils/data_utils.py:36-56: `nsamples` calibration windows of `seqlen` tokens drawn with Python's `random`
    seeded by `seed` (same draws as the reference: one randint per window, inclusive upper bound), labels masked
    except the last position, plus the tokenised test split."""
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(model, use_fast=False)
    train_ids = tok(_wikitext2_split('train'), return_tensors='pt').input_ids
    testenc = tok(_wikitext2_split('test'), return_tensors='pt')
    rng = random.Random(seed)                       # random.seed(seed) + random.randint: the same Mersenne stream
    last_start = train_ids.shape[1] - seqlen - 1
    windows = []
    for _ in range(nsamples):
        start = rng.randint(0, last_start)
        inp = train_ids[:, start:start + seqlen]
        labels = torch.full_like(inp, -100)
        labels[:, -1] = inp[:, -1]
        windows.append((inp, labels))
    return windows, testenc


def get_loaders(name, nsamples=128, seed=0, seqlen=2048, model=''):
    """utils/data_utils.py:189-204 (wikitext2 only: the hot path's headline dataset)."""
    if 'wikitext2' in name:
        return get_wikitext2(nsamples, seed, seqlen, model)
    raise ValueError("dataset %r is not available in this build (wikitext2 only)" % name)
