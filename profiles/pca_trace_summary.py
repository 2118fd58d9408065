"""Per-launch timeline of the last VB-PCA iteration from a rocprofv3 --kernel-trace CSV: name, duration, gap to the previous
launch.  usage: python profiles/pca_trace_summary.py <kernel_trace.csv> [iterations are delimited by the sweep over X: k_pca_pairs / k_pca_pass12, or k_pca_pass2 in older builds]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_pca_pairs" in r["Kernel_Name"] or "k_pca_rows" in r["Kernel_Name"]]
if len(idx) < 3:
    idx = [i for i, r in enumerate(rows) if "k_pca_pass12" in r["Kernel_Name"]]
if len(idx) < 3:                           # a build before the fused sweep
    idx = [i for i, r in enumerate(rows) if "k_pca_pass2" in r["Kernel_Name"]]
lo, hi = idx[-3] + 1, idx[-2] + 1           # one steady-state iteration: after the third-last sweep up to the second-last
prev_end = int(rows[lo - 1]["End_Timestamp"])
tot = gaps = 0.0
for r in rows[lo:hi + 8]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s %9.2f us   gap %7.2f us" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev_end) / 1e3))
    tot += (e - s) / 1e3; gaps += max(0.0, (s - prev_end) / 1e3)
    prev_end = e
print("sum of kernels %.1f us, sum of gaps %.1f us" % (tot, gaps))
