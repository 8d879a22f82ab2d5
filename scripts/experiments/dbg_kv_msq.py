import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import msq, msq.harness.gsm8k as G8
from transformers import AutoTokenizer, LlamaForCausalLM
dev = torch.device("cuda:0")
fx = "tests/golden/gsm8k_fixture"
tk = AutoTokenizer.from_pretrained(fx + "/model"); gm = LlamaForCausalLM.from_pretrained(fx + "/model", torch_dtype=torch.float16).to(dev).eval()
prompt, qs, ans = G8.load_fixture(fx)
inp = tk([prompt + "\nQuestion: " + qs[0] + "\n"], return_tensors="pt").to(dev)
with torch.no_grad():
    out = gm(**inp, use_cache=True)
pk = out.past_key_values
k = pk.layers[0].keys; v = pk.layers[0].values
print("K", tuple(k.shape), k.dtype, float(k.abs().max()), float(k.float().std()), "V", float(v.abs().max()), float(v.float().std()))
for name, fn in (("mx fp8 keys", lambda: msq.kvcache.mx_quantize_keys(k, "fp8_e4m3", 32)), ("mx fp4 keys", lambda: msq.kvcache.mx_quantize_keys(k, "fp4_e2m1", 32)),
                 ("msq keys", lambda: msq.kvcache.mx_quantize_keys(k, "fp4_e2m1", 32, 8, "fp8_e4m3")), ("msq values", lambda: msq.kvcache.mx_quantize_values(v, "fp4_e2m1", 32, 8, "fp8_e4m3")),
                 ("msq keys f32", lambda: msq.kvcache.mx_quantize_keys(k.float(), "fp4_e2m1", 32, 8, "fp8_e4m3"))):
    y = fn()
    e = (y.float() - k.float() if "keys" in name else y.float() - v.float())
    print("%-14s nan %d inf %d rel rms err %.4f max|y| %.3f" % (name, int(torch.isnan(y).sum()), int(torch.isinf(y).sum()), float(e[~torch.isnan(e)].pow(2).mean().sqrt() / k.float().std()), float(y[~torch.isnan(y)].abs().max())))
cfg = msq.kvcache.CompressionConfig(attention_number=2, streaming=True, streaming_gap=32, stream_grouping=True, compress_method="MSQ", mx_format="fp4_e2m1", mx_outlier_format="fp8_e4m3", mx_block=32).copy_for_all_attention()
acc, s = G8.evaluate(gm, tk, prompt, qs[:4], ans[:4], cfg, batch_size=4, max_new_tokens=40, return_samples=True)
print(acc, repr(s[0]["generation"][:200]))
for meth, kw in (("MX", dict(mx_format="fp4_e2m1")), ("MSQ", dict(mx_format="fp4_e2m1", mx_outlier_format="fp8_e4m3"))):
    cfg = msq.kvcache.CompressionConfig(attention_number=2, streaming=True, streaming_gap=32, stream_grouping=True, compress_method=meth, mx_block=32, **kw).copy_for_all_attention()
    cache = msq.kvcache.MXKVCache(cfg)
    with torch.no_grad():
        o = gm(**inp, past_key_values=cache, use_cache=True)
    kk = cache.layers[0].keys
    print(meth, "logits nan", int(torch.isnan(o.logits).sum()), "argmax", int(o.logits[0, -1].argmax()), tk.decode([int(o.logits[0, -1].argmax())]), "K nan", int(torch.isnan(kk).sum()), tuple(kk.shape), float((kk.float() - k.float()).abs().max()))
    vv = cache.layers[0].values
    print("   V nan", int(torch.isnan(vv).sum()), float((vv.float() - v.float()).abs().max()), "K1 nan", int(torch.isnan(cache.layers[1].keys).sum()))
k1 = pk.layers[1].keys
y = msq.kvcache.mx_quantize_keys(k1[:, :, :192].contiguous(), "fp4_e2m1", 32, 8, "fp8_e4m3")
nz = torch.isnan(y).nonzero()
print("NaN count", len(nz), "first", nz[0].tolist() if len(nz) else None)
if len(nz):
    b, h, s, d = nz[0].tolist()
    blk = k1[b, h, (s // 32) * 32:(s // 32) * 32 + 32, d].float().cpu().numpy()
    print(np.array2string(blk, precision=4, max_line_width=200))
    from oracle import oracle as O
    o = O.outlier_fakequant_lowp(blk.reshape(32, 1), "float16", 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 0, 32)
    print("oracle lowp:", o["out"].reshape(-1)[:8], "mask", o["mask"].reshape(-1).tolist(), "e_in", o["e_in"].reshape(-1), "e_out", o["e_out"].reshape(-1), "status", o["status"])
    o2 = O.outlier_fakequant(blk.reshape(32, 1), 8, 8, "fp4_e2m1", "fp8_e4m3", 2, 0, 32)
    print("oracle fp32:", o2["out"].reshape(-1)[:8], "e_in", o2["e_in"].reshape(-1), "e_out", o2["e_out"].reshape(-1))
