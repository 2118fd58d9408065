"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter values, chip clock and MFMA busy share.
usage: python profiles/pmc_summary.py <counter_collection.csv> <kernel_trace.csv>"""
import csv, sys, collections
cc, kt = sys.argv[1], sys.argv[2]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if not any(s in k for s in ("k_sweep", "k_stats", "k_prep", "k_cols", "k_gy", "k_pca_pass", "k_pca_pairs", "k_pca_rows", "_big")):
        continue
    big = [x for x in d if x > 0.5 * max(d)]
    md = sum(big) / len(big)
    print("%s  mean duration %.3f ms over %d dispatches" % (k, md * 1e3, len(big)))
    c = {n: sum(v[-len(big):]) / len(big) for n, v in acc[k].items()}
    for n, v in sorted(c.items()):
        print("   %-28s %.4g" % (n, v))
    if "GRBM_GUI_ACTIVE" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8 / md
        line = "   -> chip clock %.2f GHz" % (clk * 1e-9)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            per_simd = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
            line += "; MFMA busy per SIMD %.3g cycles = %.0f%% of the kernel's %.3g cycles" % (per_simd, 100 * per_simd / (clk * md), clk * md)
        print(line)
