"""Summarises rocprofv3 --pmc passes (counter_collection.csv) per kernel: mean counter value per dispatch.
Writes <dir>/pmc_summary.json and prints a table.  HBM traffic per launch follows
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): bytes = FETCH_SIZE*1024 (x2 on gfx950 for wide coalesced
streaming reads: this rocprofv3 tallies 128-B requests at 64 B) + WRITE_SIZE*1024."""
import csv, glob, json, os, sys, collections

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "tcnn_hip" not in name and "rocclr" not in name:
                continue
            short = name.split("tcnn_hip")[-1][:60] if "tcnn_hip" in name else name[:40]
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
summary = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, cs in summary.items():
    if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
        cs["hbm_bytes_raw"] = (cs.get("FETCH_SIZE", 0.0) + cs.get("WRITE_SIZE", 0.0)) * 1024
        cs["hbm_bytes_gfx950_streaming"] = (2 * cs.get("FETCH_SIZE", 0.0) + cs.get("WRITE_SIZE", 0.0)) * 1024
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs and cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"] > 0:
        cs["l2_hit_rate"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
json.dump(summary, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k in sorted(summary):
    print(k)
    for c in sorted(summary[k]):
        print(f"    {c:34s} {summary[k][c]:.6g}")
