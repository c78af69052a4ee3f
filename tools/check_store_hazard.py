#!/usr/bin/env python3
"""Static check for the gfx950 store-data hazard (tools/membench/store_war.hip; phx_mfma_v3common.inc: bstore_guard).

Reads AMDGPU assembly listings (`hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only x.hip -o x.s`) and
reports every buffer / global / scratch store of more than 64 bits whose data registers are written by one of the next
instructions with fewer than MIN_WAIT wait states in between (an s_nop N counts N + 1, every other instruction 1).
usage: python tools/check_store_hazard.py a.s [b.s ...]      exit code 1 if any site is found"""
import re
import sys

MIN_WAIT = 2
WIDE = re.compile(r"^(buffer|global|scratch|flat)_store_(dwordx3|dwordx4|b96|b128)\b")


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out


def check(path):
    ins = []
    for n, line in enumerate(open(path), 1):
        code = line.split(";")[0].strip()
        if not code or (code.startswith(".") and not code.endswith(":")):
            continue
        ins.append((n, code))
    bad, stores = [], 0
    for i, (n, code) in enumerate(ins):
        if not WIDE.match(code):
            continue
        stores += 1
        ops = code.split(None, 1)[1].split(",")
        # buffer_store: data is operand 0; global/scratch/flat: (addr, data, ...) -- take every vector tuple wider than one
        data = set()
        for op in ops:
            r = regs(op)
            if len(r) >= 3:
                data |= r
        ws = 0
        for n2, c2 in ins[i + 1:i + 8]:
            if c2.endswith(":"):
                break
            if c2.startswith("s_nop"):
                ws += int(c2.split()[1]) + 1
                continue
            if ws >= MIN_WAIT:
                break
            name = c2.split()[0]
            dst = regs(c2.split(None, 1)[1].split(",")[0]) if " " in c2 else set()
            # VALU writers only: the data of LDS / memory loads arrives tens of cycles after their issue
            writes_vgpr = name.startswith("v_") and not name.startswith(("v_cmp", "v_readlane", "v_readfirstlane"))
            if writes_vgpr and dst & data:
                bad.append((n, code, n2, c2, ws))
                break
            ws += 1
    return stores, bad


if __name__ == "__main__":
    rc = 0
    for path in sys.argv[1:]:
        stores, bad = check(path)
        print("%s: %d wide stores, %d with their data registers overwritten after < %d wait states" % (path, stores, len(bad), MIN_WAIT))
        for n, code, n2, c2, ws in bad[:10]:
            print("   line %d: %s\n      line %d (+%d wait states): %s" % (n, code, n2, ws, c2))
        rc |= 1 if bad else 0
    sys.exit(rc)
