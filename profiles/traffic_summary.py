"""HBM bytes per launch of the named kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) under <dir>/<COUNTER>/
t_counter_collection.csv, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: the counters are in KiB;
FETCH_SIZE reports half the bytes of wide coalesced reads (x 2: an upper bound where a kernel also issues narrow loads -- byte
masks, 8-byte operands); WRITE_SIZE is exact for 16-byte-per-lane stores.  A profiled command may launch the same kernel at
several problem sizes (bench.py: the timed workload and its small parity copy): launches are grouped by grid size and the
LARGEST grid is the one reported, the others are listed under "other_grids".
usage: python profiles/traffic_summary.py <dir> <kernel name substring> ...   -> JSON on stdout"""
import collections, csv, json, sys
out, pats = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(lambda: collections.defaultdict(dict))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("%s/%s/t_counter_collection.csv" % (out, c))):
        if r["Counter_Name"] == c:
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (k, g), v in acc.items():
        res[k][g][c] = sum(v) / len(v)
        res[k][g]["launches"] = len(v)
summary = {}
for k, grids in res.items():
    if not any(p in k for p in pats):
        continue
    def entry(v):
        rd, wr = 2.0 * v.get("FETCH_SIZE", 0.0) * 1024, v.get("WRITE_SIZE", 0.0) * 1024
        return {"fetch_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "launches_seen": v.get("launches"),
                "raw_FETCH_SIZE_KiB": v.get("FETCH_SIZE"), "raw_WRITE_SIZE_KiB": v.get("WRITE_SIZE")}
    g = max(grids)
    e = entry(grids[g])
    e["grid_size"] = g
    others = {str(o): entry(v)["hbm_bytes_per_launch"] for o, v in grids.items() if o != g}
    if others:
        e["other_grids"] = others
    summary[k] = e
json.dump(summary, sys.stdout, indent=1)
