"""A gate on the committed evidence: profiles/rNN_kernel_timings.txt (scripts/measure_kernels.py on the GPU box, one file per round) must not
show a kernel more than 15 % slower than the previous round's file unless the slowdown is acknowledged, with a reason, in
profiles/timing_acks.json.  Round 5 shipped a 2.5-2.9x regression of the strided-axis MX quantiser (a template parameter turned run-time argument
put its register tile into scratch) that its own evidence set showed and nobody read; `make check-resources` now catches the cause at build
time, this test catches the symptom whatever the cause.  CPU-only: it reads text files."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
THRESH = 1.15


def parse(path):
    """{key: [us, ...]}: a line's key is its text up to the first '<number> us' with digits of MEASURED quantities removed; every '<number> us'
    on the line is a value (position matters: 'sum / max', 'pack | unpack')."""
    out = {}
    for line in open(path):
        if line.startswith("#") or " us" not in line:
            continue
        vals = [float(v) for v in re.findall(r"([0-9]+(?:\.[0-9]+)?)\s*us\b", line)]
        m = re.search(r"[0-9]+(?:\.[0-9]+)?\s*(?:/\s*[0-9.]+\s*)?us\b", line)
        key = re.sub(r"\s+", " ", line[:m.start()]).strip(" :|")
        if key and vals:
            out.setdefault(key, vals)
    return out


def slowdowns(prev, cur):
    a, b = parse(prev), parse(cur)
    res = []
    for k in sorted(set(a) & set(b)):
        for i, (x, y) in enumerate(zip(a[k], b[k])):
            if x > 0 and y > THRESH * x:
                res.append((k, i, x, y))
    return res, len(set(a) & set(b))


def rounds():
    fs = sorted(glob.glob(os.path.join(PROF, "r[0-9][0-9]_kernel_timings.txt")))
    return [(os.path.basename(f)[:3], f) for f in fs]


def test_no_unacknowledged_slowdown_between_consecutive_rounds():
    rs = rounds()
    assert len(rs) >= 2
    acks = json.load(open(os.path.join(PROF, "timing_acks.json")))
    checked = 0
    for (ra, fa), (rb, fb) in zip(rs[:-1], rs[1:]):
        if int(rb[1:]) < 5:                       # the gate starts with round 5's file (the first one it would have caught something in)
            continue
        slow, common = slowdowns(fa, fb)
        assert common >= 30, (ra, rb, common)     # the files still describe the same kernels
        ack = acks.get(rb, {})
        for k, i, x, y in slow:
            reason = next((v for pat, v in ack.items() if pat in k), None)
            assert reason, "%s -> %s: '%s' value %d went %.1f -> %.1f us (x%.2f) and profiles/timing_acks.json has no entry for it" % (ra, rb, k, i, x, y, y / x)
        checked += 1
    assert checked >= 1


def test_the_gate_sees_round_5s_regression():
    """the mechanism on the case it was built for: r04 -> r05 lists the axis-0 tile quantiser (109.8 -> 269.7, 138.3 -> 403.3 us) and the fp32
    KV-MX keys (40.5 -> 115.7 us)"""
    slow, _ = slowdowns(os.path.join(PROF, "r04_kernel_timings.txt"), os.path.join(PROF, "r05_kernel_timings.txt"))
    keys = " | ".join(k for k, *_ in slow)
    assert "axis 0, tile 16" in keys and "axis 0, tile 32" in keys and "MX-FP8: keys" in keys
