#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 --kernel-trace --stats and --pmc passes of
bench.py, produced by scripts/profile_gpu.sh on the GPU box) into profiles/<tag>_*.{csv,json}."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda pat: sorted(glob.glob(pat), key=os.path.getmtime)[-1:]     # gpurun_out/ accumulates earlier runs
    ks = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python3 bench.py ..."}
    dom = [r for r in rows if "k_qgemm" in r["Name"] or "k_mxgemm" in r["Name"]][0]
    summary["dominant_kernel"] = dom["Name"][:100]
    summary["dominant_kernel_avg_us"] = float(dom["AverageNs"]) / 1e3
    summary["dominant_kernel_calls"] = int(dom["Calls"])
    bj = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
    # the stats average covers every launch (clock-ramp and warm-up launches included); the bench's timed region is
    # the LAST `steps` launches of the dominant kernel: average those from the kernel trace
    kt = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if kt:
        durs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                for r in csv.DictReader(open(kt[0])) if "k_qgemm" in r["Kernel_Name"] or "k_mxgemm" in r["Kernel_Name"]]
        durs.sort()
        last = [d for _, d in durs][-int(bj["steps"]):]
        summary["dominant_kernel_avg_us_timed_region"] = sum(last) / len(last) / 1e3
        summary["dominant_kernel_timed_region_launches"] = len(last)
    summary["bench_under_profiler"] = {k: bj[k] for k in ("value", "ms_per_step")}
    summary["bench_roofline_kernel_ms_under_profiler"] = bj["roofline"]["kernel_ms"]
    for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        f = newest(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if not f:
            continue
        rr = [r for r in csv.DictReader(open(f[0])) if ("k_qgemm" in r["Kernel_Name"] or "k_mxgemm" in r["Kernel_Name"]) and r["Counter_Name"] == name]
        vals = [float(r["Counter_Value"]) for r in rr]
        summary[name + "_KB_per_launch_raw"] = sum(vals) / len(vals)
        meta = rr[0]
        summary["launch"] = {k: meta[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in meta}
    if "FETCH_SIZE_KB_per_launch_raw" in summary:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE reports exactly half of the bytes of a wide coalesced
        # streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
        fetch = 2.0 * summary["FETCH_SIZE_KB_per_launch_raw"] * 1024
        write = summary["WRITE_SIZE_KB_per_launch_raw"] * 1024
        summary["traffic_bytes_per_launch"] = fetch + write
        summary["traffic_note"] = "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes); fabric-side counters, Infinity-Cache hits included"
        cfg = bj["config"]
        x_bytes = (1 + 1 / 32) if "mx" in cfg.get("layout", "") else 2          # MX path: e4m3 codes + scale bytes
        alg = cfg["M"] * cfg["K"] * x_bytes + cfg["N"] * cfg["K"] * cfg["packed_bits_per_weight"] / 8 + cfg["M"] * cfg["N"] * 2
        summary["algorithmic_bytes_per_launch"] = alg
    with open(os.path.join(dst, tag + "_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    for t in sys.argv[1:]:
        main(t)
