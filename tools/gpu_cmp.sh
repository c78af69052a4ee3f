#!/bin/bash
# Run ON THE GPU BOX: adjoint launch time of the backward-kernel variants on the same device
# usage: bash tools/gpu_cmp.sh <tag> "<workloads>" "<variants>"   variants: v1 | np2 | np4 | auto
R=$(pwd); OUT=$R/gpurun_out/${1:-cmp}; mkdir -p $OUT
for wl in ${2:-breast insilico yeast}; do
  for var in ${3:-auto v1}; do
    unset PHX_ADJ PHX_ADJ2_NP
    case $var in v1) export PHX_ADJ=v1;; np2) export PHX_ADJ=v2 PHX_ADJ2_NP=2;; np4) export PHX_ADJ=v2 PHX_ADJ2_NP=4;; esac
    timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${wl}_$var.json 2> $OUT/bench_${wl}_$var.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${wl}_$var.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$wl $var ms_per_step %.3f  fwd %.3f ms  adj %.3f ms  value %.3e" % (d["ms_per_step"], r["forward"]["launch_ms"], r["adjoint"]["launch_ms"], d["value"]))
except Exception as e:
    print("$wl $var failed:", e); print(open("$OUT/bench_${wl}_$var.err").read()[-800:])
PY
  done
done
