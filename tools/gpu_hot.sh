#!/bin/bash
# Run ON THE GPU BOX: hot (hand-pipelined) vs generic fused sweeps of k1_solve_adj2, both geometries, same device
export PHX_ADJ=v2
for wl in ${1:-breast}; do
for np in 2 4; do for sg in 0 -1; do
  export PHX_ADJ2_NP=$np PHX_STAGGER_US=$sg
  python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl np=$np', 'hot' if $sg>=0 else 'generic', 'adj %.3f ms' % d['roofline']['adjoint']['launch_ms'])"
done; done
unset PHX_ADJ2_NP PHX_STAGGER_US
PHX_ADJ=v1 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl v1 adj %.3f ms' % d['roofline']['adjoint']['launch_ms'])"
export PHX_ADJ=v2
done
