#!/usr/bin/env python3
"""GSM8K-CoT fixture of BASELINE config 4 (kv_quant/evaluation_gsm8k.py:411-533): a few-shot prompt, a set of word problems with gold answers in
GSM8K's "#### N" form, and a SMALL TRAINED Llama that solves them through a chain of thought -- all made here (no GSM8K, no prompt file
`lib_prompt/prompt_original.txt` and no Llama-2 checkpoint exist in the build image or on the GPU box).

  tests/golden/gsm8k_fixture/prompt.txt        four worked examples in the layout of the reference's prompt files ("Question: ...\\nAnswer: ... The
                                               answer is N .\\n\\n"): the evaluation prepends it to every question (:471-474)
  tests/golden/gsm8k_fixture/test.jsonl        96 problems {"question", "answer"}; the answer text ends in "#### N" like GSM8K's
  tests/golden/gsm8k_fixture/model/            LlamaForCausalLM (2 layers, hidden 128, 4 heads of 32, fp16 safetensors, < 1 MB) + WordLevel tokenizer

The problems are two-step: "<name> has A <things> , buys B more and gives C away . How many ... ?" with the chain "A + B = S . S - C = R . The answer
is R ." -- every number the chain copies (A, B, C) sits in the QUESTION, i.e. in the KV cache when the answer is generated, behind ~200 tokens of
few-shot prompt: a cache quantiser that damages those keys / values changes what is copied and the exact-match accuracy moves (2-bit groups: it
drops; 4-bit / MX-FP8: it holds), which is what config 4's metric is for.  Numbers are single tokens (0 ... 99).
The corpus, prompt, problems and tokenizer are reproduced bit for bit by re-running the script; the checkpoint is the artefact of one seeded
training run on this container's CPU cores (same quality, not the same bytes, on a re-run).  Nothing here reads /root/reference.

    python tests/golden/make_gsm8k_fixture.py            # everything (~5 min on 8 cores)
    python tests/golden/make_gsm8k_fixture.py --data-only
"""
import argparse
import json
import os
import random
import time

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "gsm8k_fixture")
MODEL = os.path.join(OUT, "model")

NAMES = ["Tom", "Ada", "Lin", "Omar", "Zoe", "Raj", "Mia", "Ben", "Eva", "Noah", "Ivy", "Leo", "Sara", "Finn", "Uma", "Kai"]
THINGS = ["apples", "pears", "books", "coins", "cards", "shells", "pens", "stamps", "marbles", "stickers", "eggs", "plums"]


def problem(rng):
    name, thing = rng.choice(NAMES), rng.choice(THINGS)
    a, b = rng.randint(2, 40), rng.randint(2, 40)
    s = a + b
    c = rng.randint(1, min(s - 1, 30))
    r = s - c
    q = "%s has %d %s , buys %d more and gives %d away . How many %s does %s have ?" % (name, a, thing, b, c, thing, name)
    chain = "%s starts with %d %s . %d + %d = %d . %d - %d = %d . The answer is %d ." % (name, a, thing, a, b, s, s, c, r, r)
    return q, chain, r


def block(q, chain):
    return "Question: %s\nAnswer: %s\n\n" % (q, chain)


def make_data():
    rng = random.Random(20260604)
    shots = [problem(rng) for _ in range(4)]
    prompt = "".join(block(q, ch) for q, ch, _ in shots).rstrip("\n")          # the evaluation adds "\nQuestion: " itself (:471-474)
    test = []
    for _ in range(96):
        q, ch, r = problem(rng)
        test.append({"question": q, "answer": "%s\n#### %d" % (ch, r)})
    train_rng = random.Random(7)
    docs = []
    for _ in range(6000):                       # a training document = 5-7 blocks back to back, as the prompt + question + generated answer look
        docs.append("".join(block(*problem(train_rng)[:2]) for _ in range(train_rng.randint(5, 7))))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "prompt.txt"), "w") as f:
        f.write(prompt)
    with open(os.path.join(OUT, "test.jsonl"), "w") as f:
        for t in test:
            f.write(json.dumps(t) + "\n")
    return prompt, test, docs


def build_tokenizer(docs):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    # a newline is a token of its own (the generation is cut at "\nQuestion: ", :515); words carry their leading blank as the Metaspace mark, so
    # that decode(encode(text)) == text exactly ("\nQuestion: " comes back without a blank behind the newline)
    pre = pre_tokenizers.Sequence([pre_tokenizers.Split("\n", behavior="isolated"), pre_tokenizers.Metaspace(replacement="\u2581", prepend_scheme="never")])
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2, "\n": 3}
    for n in range(100):
        vocab["\u2581%d" % n] = len(vocab)
    for d in docs:
        for wd, _ in pre.pre_tokenize_str(d):
            if wd not in vocab:
                vocab[wd] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre
    tok.decoder = decoders.Metaspace(replacement="\u2581", prepend_scheme="never")
    os.makedirs(MODEL, exist_ok=True)
    tok.save(os.path.join(MODEL, "tokenizer.json"))
    with open(os.path.join(MODEL, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "PreTrainedTokenizerFast", "unk_token": "<unk>", "bos_token": "<s>", "eos_token": "</s>", "pad_token": "</s>",
                   "padding_side": "left", "model_max_length": 1000000}, f, indent=1)
    return len(vocab)


def train(vocab, docs, steps, batch, lr):
    import torch
    from transformers import AutoTokenizer, LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tok = AutoTokenizer.from_pretrained(MODEL)
    ids = tok("".join(docs), return_tensors="pt").input_ids[0]
    print("train tokens %d, vocab %d" % (ids.numel(), vocab))
    V = (vocab + 63) // 64 * 64
    cfg = LlamaConfig(hidden_size=128, intermediate_size=384, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, vocab_size=V,
                      max_position_embeddings=1024, tie_word_embeddings=True, rms_norm_eps=1e-5, bos_token_id=1, eos_token_id=2, pad_token_id=2)
    m = LlamaForCausalLM(cfg)
    opt = torch.optim.AdamW(m.parameters(), lr=lr, betas=(0.9, 0.95), weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=lr, total_steps=steps, pct_start=0.05)
    g = torch.Generator().manual_seed(0)
    seqlen = 320
    t0 = time.time()
    m.train()
    for s in range(steps):
        st = torch.randint(0, ids.numel() - seqlen - 1, (batch,), generator=g)
        x = torch.stack([ids[i:i + seqlen] for i in st.tolist()])
        loss = m(x, labels=x).loss
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step(); sched.step()
        if s % 100 == 0 or s == steps - 1:
            print("step %4d loss %.4f  (%.0f s)" % (s, loss.item(), time.time() - t0), flush=True)
    m.eval()
    m.half().save_pretrained(MODEL, safe_serialization=True)
    return m, tok


def evaluate_cpu(prompt, test, n=48):
    """the reference's loop on CPU in float32, uncompressed cache (sanity of the fixture; the product loop is harness/gsm8k.py evaluate)"""
    import re
    import torch
    from transformers import AutoTokenizer, LlamaForCausalLM
    tok = AutoTokenizer.from_pretrained(MODEL)
    m = LlamaForCausalLM.from_pretrained(MODEL, torch_dtype=torch.float32).eval()
    hits = 0
    with torch.no_grad():
        for i in range(0, n, 16):
            qs = [t["question"] for t in test[i:i + 16]]
            inp = tok([prompt + "\nQuestion: " + q + "\n" for q in qs], return_tensors="pt", padding="longest")
            out = m.generate(**inp, max_new_tokens=48, do_sample=False, pad_token_id=tok.eos_token_id)
            gens = tok.batch_decode(out[:, inp.input_ids.shape[1]:], skip_special_tokens=True)
            for gtxt, t in zip(gens, test[i:i + 16]):
                cut = gtxt.split("\nQuestion: ")[0].replace(",", "")
                nums = re.findall(r"\d*\.?\d+", cut)
                gold = float(re.findall(r"\d*\.?\d+", t["answer"])[-1])
                hits += int(bool(nums) and float(nums[-1]) == gold)
            if i == 0:
                print(repr(gens[0][:160]))
    return hits / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-only", action="store_true")
    ap.add_argument("--steps", type=int, default=2500)
    args = ap.parse_args()
    prompt, test, docs = make_data()
    vocab = build_tokenizer(docs)
    if args.data_only:
        return
    train(vocab, docs, args.steps, 48, 3e-3)
    acc = evaluate_cpu(prompt, test)
    print("fp32 accuracy, uncompressed cache, 48 problems: %.3f" % acc)
    with open(os.path.join(OUT, "fixture_info.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_gsm8k_fixture.py", "steps": args.steps, "vocab": vocab, "problems": len(test),
                   "accuracy_fp32_uncompressed_first48": acc}, f, indent=1)


if __name__ == "__main__":
    main()
