#!/usr/bin/env python3
"""Condense the rocprofv3 passes of tools/collect_profiles.sh into the committed summaries under profiles/."""
import csv
import glob
import json
import os
import re
import sys

out, wl, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def short(name):
    m = re.search(r"(k1?_[a-z_0-9]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(prof, "%s_%s_kernel_stats.csv" % (tag, wl)), "w") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:25]:
            w.writerow([short(r["Name"]) if "k1" in r["Name"] or "k_" in r["Name"] else r["Name"][:90], r["Calls"],
                        r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("kernel stats:", len(rows), "kernels")

res = {}
for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = short(r["Kernel_Name"])
        if not k.startswith("k"):
            continue
        key = re.sub(r"<.*", "", k)
        a = acc.setdefault(key, [0.0, 0])
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    res["%s_KB_per_launch" % ctr] = {k: v[0] / v[1] for k, v in acc.items()}
    res["%s_launches" % ctr] = {k: v[1] for k, v in acc.items()}
if res:
    # provenance: which build these counters belong to (bench.py reports `traffic` only when the library it runs is this one)
    import hashlib
    lib = os.path.join(root, "phoenix_amd", "libphoenix_hip.so")
    if os.path.exists(lib):
        res["library_sha256"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()
    res["kernels"] = sorted(set(res.get("FETCH_SIZE_KB_per_launch", {})) | set(res.get("WRITE_SIZE_KB_per_launch", {})))
    res["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --workload %s`; "
                   "counter values are KB summed over the XCDs; FETCH_SIZE under-reports 16-B/lane streaming reads "
                   "by 2x on gfx950 (MI355X_MICROARCH.md, HBM section), bench.py applies that correction" % wl)
    json.dump(res, open(os.path.join(prof, "%s_%s_pmc_hbm.json" % (tag, wl)), "w"), indent=1)
    print(json.dumps(res, indent=1)[:1500])
