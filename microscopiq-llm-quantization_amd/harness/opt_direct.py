"""llm/opt_direct.py:51-75 opt_eval: whole-model perplexity of a model whose Linears were swapped by quantize_model
(no layer-by-layer offload on this path), and the spec the script hard-codes (:98-107)."""
import torch
import torch.nn as nn

# llm/opt_direct.py:98-105
DIRECT_MX_SPECS = {'w_elem_format': 'fp4', 'a_elem_format': 'int4', 'block_size': 128, 'custom_cuda': False,
                   'quantize_backprop': False}


@torch.no_grad()
def opt_eval(model, testenc, dev):
    print('Evaluating ...')
    model.to(dev)
    testenc = testenc.input_ids if hasattr(testenc, "input_ids") else testenc
    nsamples = testenc.numel() // model.seqlen
    nlls = []
    for i in range(nsamples):
        batch = testenc[:, (i * model.seqlen):((i + 1) * model.seqlen)].to(dev)
        lm_logits = model(batch).logits
        shift_logits = lm_logits[:, :-1, :].contiguous().float()
        shift_labels = batch[:, 1:]
        loss = nn.CrossEntropyLoss()(shift_logits.view(-1, shift_logits.size(-1)), shift_labels.reshape(-1))
        nlls.append(loss.float() * model.seqlen)
    ppl = torch.exp(torch.stack(nlls).sum() / (nsamples * model.seqlen))
    print(ppl.item())
    return ppl.item()
