#!/usr/bin/env python3
"""Build-time gate on kernel resources (`make -C csrc check-resources`, called from __graft_entry__.build()).

Reads the `-Rpass-analysis=kernel-resource-usage` remarks the Makefile writes beside every object and fails (exit 1)
when a kernel uses scratch memory (`ScratchSize [bytes/lane] > 0`) unless it is on the allow-list below WITH a reason.
Why it exists: round 5 turned a template parameter of the strided-axis MX quantiser into a run-time argument; its
register tile went to scratch (272 / 528 bytes per lane) and the kernel ran 2.5-2.9x slower without any test noticing.

Also prints, with -v, the whole table (VGPRs / AGPRs / scratch / LDS / occupancy) so that a register-pressure change is
visible in a diff of two builds."""
import re
import subprocess
import sys

# regex on the DEMANGLED kernel name -> (max scratch bytes per lane, reason).  Everything else must be 0.
ALLOW = [
    # --- generic fall-back instantiations: one lane owns a whole block of 64 / 128 elements (the harness and every BASELINE
    #     configuration use 16 or 32, which live in registers); never a measured configuration
    (r"^k_outlier_(contig|strided)<(64|128)[,>]", 1600, "block 64 / 128 fall-back, fp32"),
    (r"^k_outlier_lowp<(64|128)[,>]", 1600, "block 64 / 128 fall-back, in-dtype fp16 / bf16"),
    (r"^k_outlier_lowp_list<64[,>]", 600, "the same op-by-op path behind the packed kernels (blocks of 64: the waves they hand back)"),
    (r"^k_mx_lowp<(64|128)[,>]", 600, "block 64 / 128 fall-back, plain MX in-dtype"),
    (r"^k_act_quant<64[,>]", 300, "block 64 fall-back of the mx_ops activation quantiser (configs use 32)"),
    (r"^k_gptq_block<64>", 300, "GPTQ column solver at quant block 64 (harness: 16, BASELINE: 32)"),
    (r"^k_pack_emit<(64|128)>", 1600, "two-pass pack (posit inliers / block 128 / non-nearest rounding): block 64 / 128"),
    (r"^k_pack_tile_u<64[,>]", 300, "unified pack at quant block 64 (configs use 16 / 32)"),
    # --- the MSQ-T1 (planes) packer: the non-default layout (pack_weight falls back to it only when the unified planes are
    #     inexact); its per-lane value[64] tile is indexed by the plane loop.  Offline, once per weight; not reworked.
    (r"^k_pack_tile<", 300, "MSQ-T1 planes packer (fall-back layout, offline)"),
    # --- opt-in kernel (MSQ_GEMM_256=3), measured slower than the default rule on every whole-round grid
    (r"^k_qgemm256p<", 200, "persistent / stream-K form: opt-in only"),
    # --- 128-row (two blocks per CU, 128 + 128 registers) forms of the hand-allocated GEMMs: the spill is the fp32 epilogue's
    #     address / bias temporaries, outside the K-loop (scripts/check_isa.py asserts no scratch_ between the K-loop's barriers)
    (r"^k_qgemm256<\d, float, 8>", 140, "fp32-output epilogue of the 128-row form, outside the K-loop"),
    (r"^k_qgemm256<\d, u16, 8>", 8, "two dwords of prologue state, outside the K-loop"),
    (r"^k_mxgemm256<float, \d, 8>", 150, "fp32-output epilogue of the 128-row form, outside the K-loop"),
    (r"^k_mxgemm256<u16, \d, 8>", 16, "prologue state, outside the K-loop"),
    (r"^k_qgemm3<0, 6, (float|u16), 1, 16, 4, 1>", 20, "five dwords of prologue state (U8X extension plane pointers), outside the K-loop"),
    (r"^k_mxgemv<1, 2, 16, 4>", 12, "three dwords, decode MX-FP6 at 16 rows: epilogue"),
    # --- measured: profiles/r06_pack_occupancy_ab.txt -- the unified pack at 4 blocks per CU (128 VGPRs, these few dwords of scratch outside the
    #     per-block loops) is 5 % FASTER than at 3 blocks without scratch (151.5 against 160.1 us posit, 106.3 against 111.5 fp8)
    (r"^k_pack_tile_u<(8|16|32), (true|false), \d>", 72, "occupancy 4 with <= 72 B of scratch measured faster than occupancy 3 without (profiles/r06_pack_occupancy_ab.txt)"),
    # --- reachable only through MSQ_GEMV_U_WAVES=16 (tuning) at 33-64 rows, a regime k_qgemm_sk takes since round 6; the dispatcher's own
    #     choices for four row groups are 8 / 4 / 2 waves (no scratch)
    (r"^k_qgemv_u<\d, 4, 16, false, false>", 160, "tuning-only instantiation (MSQ_GEMV_U_WAVES=16 at 33-64 rows)"),
]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.split("\n")[:len(names)]
    except Exception:
        return names


def short(d):
    d = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
    d = re.sub(r"\(.*$", "", d)
    return d.replace("unsigned short", "u16").replace("unsigned char", "u8").replace("unsigned int", "u32")


def parse(path):
    t = open(path).read()
    rows = []
    for blk in t.split("Function Name: ")[1:]:
        name = blk.split()[0]
        def g(key):
            m = re.search(re.escape(key) + r":?\s*(\d+)", blk)
            return int(m.group(1)) if m else -1
        rows.append(dict(name=name, vgpr=g(" VGPRs"), agpr=g("AGPRs"), scratch=g("ScratchSize [bytes/lane]"),
                         lds=g("LDS Size [bytes/block]"), occ=g("Occupancy [waves/SIMD]"), sgpr_spill=g("SGPRs Spill"),
                         vgpr_spill=g("VGPRs Spill")))
    return rows


def main(argv):
    verbose = "-v" in argv
    files = [a for a in argv if not a.startswith("-")]
    if not files:
        print("usage: check_resources.py [-v] <tu>.remarks.txt ...")
        return 2
    bad, allowed, total = [], [], 0
    for f in files:
        rows = parse(f)
        if not rows:
            print("check_resources: no kernels found in %s (was the TU compiled with -Rpass-analysis=kernel-resource-usage?)" % f)
            return 1
        dn = demangle([r["name"] for r in rows])
        seen = set()
        for r, d in zip(rows, dn):
            if r["name"] in seen:
                continue
            seen.add(r["name"])
            total += 1
            s = short(d)
            if verbose:
                print("%-28s %-90s vgpr %3d agpr %3d scratch %4d lds %6d occ %2d" % (f.split("/")[-1][:-12], s[:90], r["vgpr"], r["agpr"], r["scratch"], r["lds"], r["occ"]))
            if r["scratch"] > 0:
                ok = None
                for pat, lim, why in ALLOW:
                    if re.search(pat, s) and r["scratch"] <= lim:
                        ok = why
                        break
                (allowed if ok else bad).append((f.split("/")[-1], s, r["scratch"], r["vgpr"], ok))
    for f, s, sc, vg, why in allowed:
        print("allowed  %-26s %-80s scratch %4d B/lane  (%s)" % (f, s[:80], sc, why))
    for f, s, sc, vg, _ in bad:
        print("SCRATCH  %-26s %-80s scratch %4d B/lane, %d VGPRs" % (f, s[:80], sc, vg))
    print("check_resources: %d kernels, %d on the allow-list, %d using scratch outside it" % (total, len(allowed), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
