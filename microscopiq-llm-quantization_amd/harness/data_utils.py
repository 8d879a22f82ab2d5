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


def wikitext2_rows(split, data_dir=None):
    """The `text` column of wikitext-2-raw-v1 for `split`.  Without network the dataset has to be on local disk:
    ``data_dir`` (or $MSQ_WIKITEXT2_DIR) may hold the raw distribution (wiki.<split>.raw, one row per line, line ends
    kept), or a HuggingFace `save_to_disk` / parquet copy of the dataset; otherwise the hub name is tried."""
    import os
    data_dir = data_dir or os.environ.get("MSQ_WIKITEXT2_DIR")
    if data_dir:
        raw = os.path.join(data_dir, "wiki.%s.raw" % split)
        if os.path.exists(raw):
            with open(raw, encoding="utf-8") as f:
                return f.read().splitlines(keepends=True)
        from datasets import load_dataset, load_from_disk
        if os.path.exists(os.path.join(data_dir, "dataset_dict.json")):
            return load_from_disk(data_dir)[split]['text']
        pq = [os.path.join(r, fn) for r, _, fs in os.walk(data_dir) for fn in fs if fn.endswith(".parquet") and split in fn]
        if pq:
            return load_dataset("parquet", data_files={split: sorted(pq)}, split=split)['text']
        raise FileNotFoundError("no wiki.%s.raw, saved dataset or %s*.parquet under %s" % (split, split, data_dir))
    from datasets import load_dataset
    return load_dataset('wikitext', 'wikitext-2-raw-v1', split=split)['text']


def get_wikitext2(nsamples, seed, seqlen, model, data_dir=None, tokenizer=None):
    """utils/data_utils.py:36-56: the whole split joined with "\n\n" and tokenised ONCE; `nsamples` calibration windows
    of `seqlen` tokens drawn with Python's `random` seeded by `seed` (one randint per window, inclusive upper bound
    len - seqlen - 1), labels -100 except the last position; plus the tokenised test split."""
    if tokenizer is None:
        from transformers import AutoTokenizer
        tokenizer = AutoTokenizer.from_pretrained(model, use_fast=False)
    train_ids = tokenizer("\n\n".join(wikitext2_rows('train', data_dir)), return_tensors='pt').input_ids
    testenc = tokenizer("\n\n".join(wikitext2_rows('test', data_dir)), return_tensors='pt')
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


def get_loaders(name, nsamples=128, seed=0, seqlen=2048, model='', data_dir=None):
    """utils/data_utils.py:189-204 (wikitext2 only: the hot path's headline dataset)."""
    if 'wikitext2' in name:
        return get_wikitext2(nsamples, seed, seqlen, model, data_dir=data_dir)
    raise ValueError("dataset %r is not available in this build (wikitext2 only)" % name)
