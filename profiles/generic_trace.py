"""Per-launch durations of the generic path from a rocprofv3 --kernel-trace CSV, grouped by grid size (blocks) and LDS bytes:
python profiles/generic_trace.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    if "k_tape" in r["Kernel_Name"]:
        by[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"]) // 256, r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(by.items()):
    v.sort()
    print("%-14s blocks %5d lds %8s: n %6d  median %9.1f us  max %9.1f  total %8.1f ms" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[-1], sum(v) / 1e3))
