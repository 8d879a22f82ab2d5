"""GSM8K-CoT exact-match scoring (kv_quant/evaluation_gsm8k.py:63-85 evaluate_pred_answer, accuracy :531-533): the
metric of BASELINE config 4.  Pure host logic; pinned by KATs produced with the reference's own function
(tests/golden/gsm8k_scorer.json)."""
import re

_NUMBER = r"\d*\.?\d+"


def evaluate_pred_answer(pred_str, ans_str):
    """Last number of the generation against the last number of the gold answer (commas stripped, compared as floats).
    Returns (is_pred_true, pred, pred_list, gold, gold_list) like the reference."""
    pred_str, ans_str = pred_str.replace(",", ""), ans_str.replace(",", "")
    pred_list = re.findall(_NUMBER, pred_str)
    gold_list = re.findall(_NUMBER, ans_str)
    gold = float(gold_list[-1])
    if len(pred_list) >= 1:
        pred = float(pred_list[-1])
        return pred == gold, pred, pred_list, gold, gold_list
    return False, None, pred_list, gold, gold_list


def accuracy(generations, answers, generation_split="\nQuestion: "):
    """evaluation_gsm8k.py:515-533: the generation is cut at `generation_split` (the start of the next few-shot block)
    before scoring; accuracy = mean of the exact matches."""
    hits = [evaluate_pred_answer(g.split(generation_split)[0], a)[0] for g, a in zip(generations, answers)]
    return sum(hits) / len(hits) if hits else 0.0


def evaluate(model, tokenizer, prompt_cot, questions, answers, cache_config=None, batch_size=8, max_new_tokens=256, max_length=None,
             generation_split="\nQuestion: ", zero_shot=False, device=None, return_samples=False):
    """The evaluation loop of kv_quant/evaluation_gsm8k.py:455-533 on a model + tokenizer the caller has loaded: per batch the few-shot prompt
    is prepended to every question (:471-474; `zero_shot` swaps in the reference's one-line instruction, :465-470), the batch is tokenised with
    left padding ("longest", truncation on: :476-481), generated greedily with the cache in use (:483-503: do_sample False, temperature / top_k /
    top_p None, pad = eos), the new tokens are decoded (:504-507), every generation is cut at `generation_split` -- the start of the next few-shot
    block the model goes on to write -- and its last number is compared with the last number of the gold answer (:515-517); accuracy = mean of the
    exact matches (:531-533).

    `cache_config`: a kvcache.CompressionConfig (already `copy_for_all_attention()`-ed, as :405-407 does) -> every `generate` call gets a fresh
    ``MXKVCache(cache_config)`` whose layers fake-quantise K / V at the GEAR hook on the GPU (the reference passes `compress_config` to its
    SimulatedGear* model class, which owns the hook: modeling_llama_new.py:944-1030); None -> the model's own dynamic cache (the fp16 baseline).
    Returns the accuracy, or (accuracy, samples) with one dict per problem (the fields of the reference's EvaluationSample, :518-527)."""
    import torch
    from ..kvcache import MXKVCache
    dev = device if device is not None else next(model.parameters()).device
    if zero_shot:
        prompt_cot = "answer the question through the form of The answer is xxx. Do not generate others."
    samples = []
    with torch.no_grad():
        for i in range(0, len(questions), batch_size):
            qs, gold = list(questions[i:i + batch_size]), list(answers[i:i + batch_size])
            prompts = [prompt_cot + "\nQuestion: " + q + "\n" for q in qs]
            inputs = tokenizer(prompts, return_tensors="pt", padding="longest", truncation=True).to(dev)
            kw = dict(return_dict_in_generate=True, max_new_tokens=max_new_tokens, output_scores=False, pad_token_id=tokenizer.eos_token_id,
                      use_cache=True, do_sample=False, temperature=None, top_k=None, top_p=None)
            if max_length is not None:
                kw["max_length"] = max_length
            if cache_config is not None:
                kw["past_key_values"] = MXKVCache(cache_config)
            inputs.pop("token_type_ids", None)
            out = model.generate(**inputs, **kw)
            gens = tokenizer.batch_decode(out.sequences[:, inputs.input_ids.shape[1]:], skip_special_tokens=True)
            for q, g, a in zip(qs, gens, gold):
                ok, pred, pred_list, label, gold_list = evaluate_pred_answer(g.split(generation_split)[0], a)
                samples.append({"question": q, "generation": g, "answer": a, "list_from_pred": pred_list, "list_from_answer": gold_list,
                                "pred": pred, "label": label, "is_pred_true": bool(ok)})
    acc = sum(s["is_pred_true"] for s in samples) / len(samples) if samples else 0.0
    return (acc, samples) if return_samples else acc


def load_fixture(root):
    """(prompt, questions, answers) of a GSM8K-style fixture directory: prompt.txt (the few-shot block, as lib_prompt/<prompt_file>, :455-456) and
    test.jsonl with GSM8K's {"question", "answer"} rows."""
    import json
    import os
    with open(os.path.join(root, "prompt.txt")) as f:
        prompt = f.read()
    qs, ans = [], []
    with open(os.path.join(root, "test.jsonl")) as f:
        for line in f:
            if line.strip():
                r = json.loads(line)
                qs.append(r["question"]); ans.append(r["answer"])
    return prompt, qs, ans
