#!/bin/bash
# usage: tools/build_one.sh phx_adj3 [phx_fwd3 ...]   -- compile the named units only, relink, mark the rest fresh
cd /root/repo/phoenix_amd/csrc
FLAGS=$(python -c "import sys; sys.path.insert(0,'/root/repo'); from phoenix_amd import build as b; print(' '.join(b.FLAGS))")
for u in "$@"; do
  UF=$(python -c "import sys; sys.path.insert(0,'/root/repo'); from phoenix_amd import build as b; print(' '.join(b.UNIT_FLAGS.get('$u.hip', [])))")
  ( /opt/rocm/bin/hipcc -save-temps=obj $FLAGS $UF -c $u.hip -o _obj/$u.o > /tmp/build_$u.log 2>&1; echo "$u rc=$?" ) &
done
wait
for u in "$@"; do grep -i "error" /tmp/build_$u.log | head -5; f=$(ls _obj/$u-hip-amdgcn-amd-amdhsa-gfx950.s 2>/dev/null); [ -n "$f" ] && mv $f _obj/$u.s; rm -f _obj/$u-hip-* _obj/$u-host-* ; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libphoenix_hip.so _obj/phx_adj2.o _obj/phx_adj3.o _obj/phx_adj3c.o _obj/phx_engine.o _obj/phx_fwd3.o _obj/phx_fwd3c.o _obj/phx_v1.o
# (only the library is marked fresh: objects of other units that a header edit made stale stay stale, so that
#  `python -m phoenix_amd.build` rebuilds them -- a library linked here may mix objects of two header versions)
touch ../libphoenix_hip.so
