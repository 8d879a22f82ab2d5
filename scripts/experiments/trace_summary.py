"""per (kernel, grid) duration summary of a rocprofv3 --kernel-trace csv: median of the first and of the second half of the calls in time order"""
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        d[((re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"]) or re.search(r"\w+", r["Kernel_Name"])).group(0), r["Grid_Size_X"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(d.items()):
    v.sort(); n = len(v); a = sorted(x[1] for x in v[:n // 2] or v); b = sorted(x[1] for x in v[n // 2:])
    print("%-50s grid %9s n %4d  median 1st half %8.1f us  2nd half %8.1f us" % (k[0], k[1], n, a[len(a) // 2] / 1e3, b[len(b) // 2] / 1e3))
