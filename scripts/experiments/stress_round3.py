"""Round 3, second half: run-to-run identity of the kernels added there, half of the repetitions beside a bandwidth hog on a second
stream: the wide-projection decode kernel (sixteen- and eight-wave blocks, 1 / 2 row groups, fp16 and bf16 activations), the MX decode
kernel's eight-wave blocks, the register-resident LayerNorm / gelu, the division-free KV group quantiser, the half-precision MX
quantiser.  REPS launches per case must equal the first one bit for bit."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import msq
from msq import qlinear, kvcache, vector_ops
dev = torch.device("cuda:0"); torch.manual_seed(6)
REPS = int(os.environ.get("REPS", 300))
hog_a = torch.empty(64 << 20, dtype=torch.float32, device=dev); hog_b = torch.empty_like(hog_a)
side = torch.cuda.Stream()
bad = 0


def same(a, b):
    return ((a == b) | (a != a) & (b != b)).all()


def run(name, fn):
    global bad
    y0 = fn()
    d = torch.zeros((), dtype=torch.int64, device=dev)
    for r in range(REPS):
        if r % 2:
            with torch.cuda.stream(side):
                hog_b.copy_(hog_a)
        d += (~same(fn(), y0)).to(torch.int64)
    torch.cuda.synchronize()
    n = int(d.item()); bad += n
    print("%-70s %d launches, %d differing" % (name, REPS, n), flush=True)


for (N, K) in ((12288, 4096), (22016, 4096), (16640, 1024)):
    W = torch.randn(N, K, device=dev) * 0.02; W[torch.rand(N, K, device=dev) < 0.005] *= 16
    for fo in ("posit8_es1", "fp8_e4m3"):
        P = qlinear.pack_weight(W, 8, 8, "fp4_e2m1", fo, 2, 32, layout="unified")
        for M in (1, 16, 32):
            for dt in (torch.bfloat16, torch.float16):
                x = torch.randn(M, K, device=dev).to(dt)
                run("decode N%d K%d %s M%d %s" % (N, K, fo, M, str(dt)[6:]), lambda: qlinear.qlinear(x, P, None, torch.float32))
    P8 = qlinear.mx_pack_values(msq.quant.outlier_fakequant(W, 8, 8, "fp4_e2m1", "fp8_e4m3", 2, -1, 32)["out"])
    for M in (1, 16):
        x = torch.randn(M, K, device=dev)
        run("MX decode N%d K%d e4m3 operand M%d" % (N, K, M), lambda: qlinear.qlinear_mx_w4a8(x, P8, None, torch.float32))
    del W
sp = msq.specs.finalize_mx_specs({"w_elem_format": "fp6_e3m2", "a_elem_format": "fp6_e3m2", "scale_bits": 4, "block_size": 32, "bfloat": 16, "custom_cuda": True})
X = torch.randn(2048, 4096, device=dev); w = torch.randn(4096, device=dev); b = torch.randn(4096, device=dev)
run("LayerNorm [2048, 4096]", lambda: vector_ops.layer_norm(X, w, b, 1e-12, sp))
run("LayerNorm [300, 8192]", lambda: vector_ops.layer_norm(X.reshape(-1, 8192)[:300].contiguous(), torch.cat([w, w]), torch.cat([b, b]), 1e-5, sp))
run("gelu", lambda: vector_ops.gelu(X, mx_specs=sp))
for dt in (torch.float16, torch.float32):
    C = torch.randn(1, 32, 4096, 128, device=dev).to(dt)
    for gs, along in ((4096, False), (32, False), (32, True), (128, True)):
        f = kvcache.fake_groupwise_channel_asymmetric_quantization_new if along else kvcache.fake_groupwise_token_asymmetric_quantization
        run("KV %s groups of %d %s" % (str(dt)[6:], gs, "tokens" if along else "head.dim"), lambda: f(C, 4, gs))
    if dt == torch.float16:
        run("KV MX keys fp16", lambda: kvcache.mx_quantize_keys(C, "fp8_e4m3", 32))
        run("KV MX values fp16", lambda: kvcache.mx_quantize_values(C, "fp8_e4m3", 32))
print("TOTAL bad:", bad)
sys.exit(1 if bad else 0)
