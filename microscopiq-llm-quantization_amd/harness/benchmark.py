"""Per-token latency of a (quantised, packed) causal LM, the harness's ``benchmark`` (llm/opt.py:332-376): feed the
prompt one token at a time with the KV cache of the previous step, synchronise after every token, print every step's
seconds and the median; ``check=True`` also accumulates the next-token loss and prints the perplexity.

The reference clears each layer's cache entry through forward hooks on ``model.model.decoder.layers`` (a memory measure for
its ``opt_multigpu`` placement, :296-330, which spreads the layers of one process over several GPUs).  Here a model lives on
one MI355X (288 GB) or is sharded row-parallel with one process per GPU (qlinear.RowParallelQuantLinear), so there is no
placement step and the cache is simply replaced by the step's new one.
"""
import time

import numpy as np
import torch
import torch.nn as nn


def benchmark(model, input_ids, check=False, dev=None, log=print, skip=0):
    """Returns {"times": [...], "median": s, "ppl": float or None}.  ``skip``: leading steps left out of the median (0 =
    the reference's figure; the first steps of a process include allocator and clock warm-up)."""
    if dev is None:
        dev = next(model.parameters()).device
    if not torch.device(dev).type == "cuda":
        raise RuntimeError("benchmark needs the model on a CUDA/HIP device")
    input_ids = input_ids.to(dev)
    torch.cuda.synchronize(dev)
    log('Benchmarking ...')
    loss = nn.CrossEntropyLoss()
    tot = 0.
    past = None
    times = []
    n = input_ids.numel()
    with torch.no_grad():
        attention_mask = torch.ones((1, n), device=dev)
        for i in range(n):
            tick = time.time()
            out = model(input_ids[:, i].reshape((1, -1)), past_key_values=past, use_cache=True,
                        attention_mask=attention_mask[:, :(i + 1)].reshape((1, -1)))
            torch.cuda.synchronize(dev)
            times.append(time.time() - tick)
            log(i, times[-1])
            if check and i != n - 1:
                tot += loss(out.logits[0].to(dev), input_ids[:, (i + 1)].to(dev)).float()
            past = out.past_key_values
            del out
        torch.cuda.synchronize(dev)
    med = float(np.median(times[skip:] if len(times) > skip else times))
    log('Median:', med)
    ppl = None
    if check and n > 1:
        ppl = torch.exp(tot / (n - 1)).item()
        log('PPL:', ppl)
    return {"times": times, "median": med, "ppl": ppl}
