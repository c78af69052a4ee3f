#!/usr/bin/env python3
"""Diagnostic: registers / scratch / occupancy of every kernel in phx_engine.hip (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "phoenix_amd", "csrc", "phx_engine.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                      "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_phx_res.o"] + sys.argv[1:],
                     capture_output=True, text=True).stderr
cur, rows = None, []
for l in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(": ")[1]}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = r["name"]
    m = re.search(r"\d+(k1?_[a-z_]+)(I[^E]*E)?", n)
    tag = (m.group(1) + (m.group(2) or "")) if m else n[:40]
    print("%-30s V=%-4s A=%-4s scratch=%-5s occ=%s sgpr_spill=%s vgpr_spill=%s lds=%s" % (
        tag, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
        r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]")))
