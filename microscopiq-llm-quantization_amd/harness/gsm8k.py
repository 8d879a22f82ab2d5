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
