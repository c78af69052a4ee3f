"""Builds libphoenix_hip.so (the C-ABI engine, include/phoenix_hip.h) in-tree with hipcc for gfx950.

Every `.hip` file under csrc/ is one translation unit; they are compiled in parallel into csrc/_obj/ and linked
into one shared library.  A unit is recompiled when it, any `.inc` / `.hpp` under csrc/ or the public header is
newer than its object (the units include each other's pieces, so dependencies are tracked per directory)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libphoenix_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Per-unit flags.  The third-generation solve kernels are built WITHOUT inter-procedural register allocation: sibling
# instantiations of the same source (an eight-wave forward form, a run-time gene-block split) computed wrong dopri5 step
# sizes on lanes {12-15, 28-31, 44-47, 60-63} of the controller wave with IPRA on and right ones with it off (DESIGN.md
# section 2, "one signature").  The committed instantiations are bit-identical under both settings, so the flag costs
# nothing and takes the shipped kernels out of the reach of that register-allocation pattern;
# tests/test_gpu_parity.py::test_step_counts_of_the_third_generation_kernels_match_the_first names a recurrence.
UNIT_FLAGS = {"phx_fwd3.hip": ["-mllvm", "-enable-ipra=false"], "phx_adj3.hip": ["-mllvm", "-enable-ipra=false"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    """every file a translation unit may include"""
    return sorted(glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.hpp")) +
                  [os.path.join(ROOT, "include", "phoenix_hip.h")])


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _obj_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(src):
    o = _obj_of(src)
    if not os.path.exists(o):
        return True
    ot = os.path.getmtime(o)
    return any(os.path.getmtime(p) > ot for p in [src] + deps())


def needs_build():
    if not os.path.exists(LIB):
        return True
    lt = os.path.getmtime(LIB)
    return any(_stale(s) or os.path.getmtime(_obj_of(s)) > lt for s in sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in sources() if force or _stale(s)]
    procs = []
    for s in todo:
        cmd = [hipcc()] + FLAGS + UNIT_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", _obj_of(s)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj_of(s) for s in sources()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
