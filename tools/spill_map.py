"""Attributes register spills of a kernel to source regions: counts scratch_load / scratch_store / v_writelane /
v_readlane / v_mfma between the `; PHXMARK <name>` comments the kernels emit through inline asm.
usage: python tools/spill_map.py file.s"""
import collections
import re
import sys

cur = "(prologue)"
order = []
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open(sys.argv[1]):
    m = re.search(r"; PHXMARK (.*)", line)
    if m:
        cur = m.group(1).strip()
        if cur not in order:
            order.append(cur)
        continue
    t = line.strip().split(" ")[0]
    if t.startswith("scratch_load"):
        cnt[cur]["sld"] += 1
    elif t.startswith("scratch_store"):
        cnt[cur]["sst"] += 1
    elif t.startswith("v_writelane"):
        cnt[cur]["wl"] += 1
    elif t.startswith("v_readlane"):
        cnt[cur]["rl"] += 1
    elif t.startswith("v_mfma"):
        cnt[cur]["mfma"] += 1
    elif t.startswith("v_accvgpr"):
        cnt[cur]["acc"] += 1
    elif t.startswith("v_") or t.startswith("ds_") or t.startswith("global_") or t.startswith("s_"):
        cnt[cur]["other"] += 1
print("%-52s %6s %6s %6s %6s %6s %6s %7s" % ("region", "mfma", "sld", "sst", "wlane", "rlane", "accmv", "other"))
for k in ["(prologue)"] + order:
    c = cnt[k]
    print("%-52s %6d %6d %6d %6d %6d %6d %7d" % (k, c["mfma"], c["sld"], c["sst"], c["wl"], c["rl"], c["acc"], c["other"]))
