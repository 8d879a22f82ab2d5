#!/usr/bin/env python3
"""Static check of the GEMM kernels' ISA (run after editing csrc/msq_gemm.hip): compiles the device code to assembly and
reports, per kernel, waterfall loops (s_cbranch_execnz beyond the float epilogue's), scratch use, readfirstlane and
quarter-rate v_mul_lo counts.  Exit code 1 if a bf16-output GEMM kernel has a waterfall loop or touches scratch."""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(HERE, "..", "microscopiq-llm-quantization_amd", "csrc", "msq_gemm.hip")
out = os.path.join(tempfile.gettempdir(), "msq_gemm_check.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
                       "--cuda-device-only", "-S", src, "-o", out] + sys.argv[1:], stderr=subprocess.DEVNULL)
s = open(out).read()
bad = 0
for nm in re.findall(r'^(_Z\d+k_(?:qgemm3|mxgemm|qgemv|mxgemv)\S*):', s, re.M):
    i = s.index('\n' + nm + ':'); j = s.index('s_endpgm', i)
    body = s[i:j]
    # waterfall loops are BACKWARD s_cbranch_execnz branches (the label is defined before the branch); forward ones are plain
    # control flow (e.g. the block-order selection in the prologue)
    w = 0
    for mm in re.finditer(r's_cbranch_execnz (\S+)', body):
        lab = body.find('\n' + mm.group(1) + ':')
        if lab != -1 and lab < mm.start():
            w += 1
    sc = body.count('scratch_')
    line = "%-62s execnz %3d scratch %3d readfirstlane %3d v_mul_lo %3d valu %5d" % (nm[3:65], w, sc, body.count('readfirstlane'), body.count('v_mul_lo'), len(re.findall(r'\n\s+v_(?!mfma)', body)))
    main = ('k_qgemm3' in nm or 'k_mxgemm' in nm)
    flag = main and ((w and ('Et' in nm.split('Li')[2] if 'qgemm3' in nm else 'ItL' in nm)) or sc > 2)
    if flag:
        bad += 1
    print(("!! " if flag else "   ") + line)
# vmcnt accounting of the K-loops (k_qgemm3, k_mxgemm): between two barriers a step issues L vector-memory loads and ends with
# s_waitcnt vmcnt(N); N > L lets loads of an EARLIER step stay in flight across the barrier -- only harmless when the next
# step multiplies nothing (the drain before the epilogue).  (hipcc deletes dead loads of tail steps: see DESIGN.md 5.0.)
for nm in re.findall(r'^(_Z\d+k_(?:qgemm3|mxgemm)\S*):', s, re.M):
    i = s.index('\n' + nm + ':'); j = s.index('s_endpgm', i)
    segs, cur = [], {"loads": 0, "wait": None, "mfma": 0}
    for l in s[i:j].split('\n'):
        l = l.strip()
        if l.startswith('buffer_load') or l.startswith('global_load'):
            cur["loads"] += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l:
            cur["wait"] = int(re.search(r'vmcnt\((\d+)\)', l).group(1))
        elif l.startswith('v_mfma'):
            cur["mfma"] += 1
        elif l.startswith('s_barrier'):
            segs.append(cur); cur = {"loads": 0, "wait": None, "mfma": 0}
    segs.append(cur)
    for k in range(1, len(segs) - 1):
        a, b = segs[k], segs[k + 1]
        if a["wait"] is not None and a["wait"] > a["loads"] and b["mfma"] > 0 and a["mfma"] > 0:
            print("!! %-62s K-step segment %d: %d loads issued but vmcnt(%d) before the barrier, and the next segment multiplies" % (nm[3:65], k, a["loads"], a["wait"]))
            bad += 1
sys.exit(1 if bad else 0)
