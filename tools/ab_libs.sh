#!/bin/bash
# Run ON THE GPU BOX: bench lines (ms_per_step, forward / backward launch) of several builds of the library on the same
# device:  bash tools/ab_libs.sh hip variant1 variant2 ...   (phoenix_amd/libphoenix_<name>.so; PHX_DIAG=1 PHX_LIB=)
for lib in "$@"; do
  PHX_DIAG=1 PHX_LIB=$PWD/phoenix_amd/libphoenix_$lib.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s step %.4f ms  fwd %.4f  adj %.4f' % ('$lib', d['ms_per_step'], d['roofline']['forward']['launch_ms'], d['roofline']['launch_ms']))"
done
