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
# Per-unit flags.  phx_fwd3.hip: the iterative-ILP machine scheduler (round 4, same-device A/B: forward launch 0.2413 ->
# 0.2394 ms; the same flag makes k1_solve_adj3 slower, 0.536 -> 0.55 ms, so that unit keeps the default) and phx_adj2.hip
# (C3 yeast backward 7.28 -> 6.86 ms, C2 in-silico 0.5415 -> 0.5367 ms).  phx_adj3.hip was also tried with max-ilp,
# max-memory-clause, iterative-minreg, iterative-maxocc (all slower: 0.545 ... 0.580 ms), -fno-slp-vectorize, -O2 and the
# schedule-metric-bias / relaxed-occupancy knobs (no difference); phx_engine.hip with iterative-ilp crashes this clang
# (segmentation fault in the backend), so it keeps the default -- its first-generation solve kernels were moved to
# phx_v1.hip for the flag (C5 step 47.9 -> 47.5 ms, C2 forward 0.213 -> 0.205 ms; phx_engine.hip now compiles in 40 s).  Round 3 suspected inter-procedural register allocation behind the "trajectories
# 12..15 take thousands of steps" signature and round 4 first built the third-generation kernels with
# `-mllvm -enable-ipra=false`; the cause turned out to be a gfx950 store-data hazard the compiler does not cover
# (tools/membench/store_war.hip), fixed in the source (phx_mfma_v3common.inc: bstore_guard) and checked statically after
# every build by tools/check_store_hazard.py (tests/test_abi_cpu.py runs it on the listings build() leaves in _obj/).
UNIT_FLAGS = {"phx_fwd3.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
              "phx_adj2.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
              "phx_v1.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}
# Diagnostic build (PHX_PROF=2: per-block timers inside the sweeps of the third-generation kernels): the marks are compiled
# in only with -DPHX_PROF_BLOCKS -- as run-time branches they split the sweep body into several scheduling regions and cost
# 2.3 % (forward) / 1.5 % (backward) of the third-generation launches and 3.3 % of the second backward kernel at C3 (round 4).  build_prof() writes libphoenix_prof.so next to the library.
PROF_LIB = os.path.join(HERE, "libphoenix_prof.so")
PROF_UNITS = ("phx_fwd3.hip", "phx_adj3.hip", "phx_adj2.hip")
# Units whose device assembly is also written to csrc/_obj/<unit>.s: every kernel that stores through buffer resources
# (the store-data hazard is checked on these listings, tools/check_store_hazard.py)
LISTINGS = ("phx_fwd3.hip", "phx_adj3.hip", "phx_fwd3c.hip", "phx_adj3c.hip")


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


def listing_of(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".s")


def _keep_listing(src, dst, stem=None):
    """moves the gfx950 assembly `-save-temps=obj` left beside the object to `dst` and deletes the other temporaries"""
    # (the temporaries are named after the SOURCE file, whatever the object is called)
    stem = os.path.basename(src)[:-4]
    found = False
    for f in (glob.glob(os.path.join(OBJ, stem + "-hip-*")) + glob.glob(os.path.join(OBJ, stem + "-host-*")) +
              glob.glob(os.path.join(OBJ, stem + ".hip-hip-*"))):
        if f.endswith("gfx950.s"):
            os.replace(f, dst)
            found = True
        else:
            os.remove(f)
    if not found:
        raise RuntimeError("no device assembly was kept for %s" % src)


def _stale(src):
    o = _obj_of(src)
    if not os.path.exists(o) or (os.path.basename(src) in LISTINGS and not os.path.exists(listing_of(src))):
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
        # the checked listing is the device assembly of THIS invocation (-save-temps keeps what the object is assembled
        # from), not of a second compile whose flags could drift from the first
        if os.path.basename(s) in LISTINGS:
            cmd.insert(1, "-save-temps=obj")
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for s in todo:
        if os.path.basename(s) in LISTINGS:
            _keep_listing(s, listing_of(s))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [_obj_of(s) for s in sources()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def build_prof(verbose=False, check_stale=True):
    """libphoenix_prof.so: the library with the per-block timer marks compiled into the second / third-generation kernels
    (tools/prof_segments.py loads it for PHX_PROF=2); the other units are taken from the regular build.
    check_stale=False: an existing file is used as it is (on a GPU box the snapshot's modification times are not those
    of the build container, and a rebuild there costs four minutes of GPU time)."""
    if os.path.exists(PROF_LIB) and not check_stale:
        return PROF_LIB
    build()
    stale = not os.path.exists(PROF_LIB) or any(os.path.getmtime(p) > os.path.getmtime(PROF_LIB) for p in sources() + deps())
    if not stale:
        return PROF_LIB
    procs, objs = [], []
    for s in sources():
        b = os.path.basename(s)
        if b in PROF_UNITS:
            o = os.path.join(OBJ, b[:-4] + ".prof.o")
            cmd = [hipcc()] + FLAGS + UNIT_FLAGS.get(b, []) + ["-DPHX_PROF_BLOCKS", "-c", s, "-o", o]
            if b in LISTINGS:
                cmd.insert(1, "-save-temps=obj")
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
            objs.append(o)
        else:
            objs.append(_obj_of(s))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for s in sources():
        b = os.path.basename(s)
        if b in PROF_UNITS and b in LISTINGS:      # the diagnostic build's listings are checked too (prof_listings())
            _keep_listing(s, os.path.join(OBJ, b[:-4] + ".prof.s"))
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROF_LIB] + objs, cwd=CSRC)
    return PROF_LIB


def prof_listings():
    return [os.path.join(OBJ, b[:-4] + ".prof.s") for b in PROF_UNITS if b in LISTINGS]


if __name__ == "__main__":
    import sys
    if "--prof" in sys.argv:
        print(build_prof(verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
