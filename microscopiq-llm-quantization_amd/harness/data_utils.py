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


def get_wikitext2(nsamples, seed, seqlen, model):
    """utils/data_utils.py:36-56: (trainloader of nsamples random seqlen windows, testenc)."""
    from datasets import load_dataset
    from transformers import AutoTokenizer
    traindata = load_dataset('wikitext', 'wikitext-2-raw-v1', split='train')
    testdata = load_dataset('wikitext', 'wikitext-2-raw-v1', split='test')
    tokenizer = AutoTokenizer.from_pretrained(model, use_fast=False)
    trainenc = tokenizer("\n\n".join(traindata['text']), return_tensors='pt')
    testenc = tokenizer("\n\n".join(testdata['text']), return_tensors='pt')
    random.seed(seed)
    trainloader = []
    for _ in range(nsamples):
        i = random.randint(0, trainenc.input_ids.shape[1] - seqlen - 1)
        inp = trainenc.input_ids[:, i:i + seqlen]
        tar = inp.clone()
        tar[:, :-1] = -100
        trainloader.append((inp, tar))
    return trainloader, testenc


def get_loaders(name, nsamples=128, seed=0, seqlen=2048, model=''):
    """utils/data_utils.py:189-204 (wikitext2 only: the hot path's headline dataset)."""
    if 'wikitext2' in name:
        return get_wikitext2(nsamples, seed, seqlen, model)
    raise ValueError("dataset %r is not available in this build (wikitext2 only)" % name)
