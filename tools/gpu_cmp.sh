#!/bin/bash
# Run ON THE GPU BOX: adjoint launch time of the two generations of the backward kernel on every workload
R=$(pwd); OUT=$R/gpurun_out/${1:-cmp}; mkdir -p $OUT
for wl in breast insilico yeast; do
  for adj in v2 v1; do
    if [ $adj = v1 ]; then export PHX_ADJ=v1; else unset PHX_ADJ; fi
    timeout 600 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_${wl}_$adj.json 2> $OUT/bench_${wl}_$adj.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${wl}_$adj.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$wl adj=$adj ms_per_step %.3f  fwd %.3f ms  adj %.3f ms  value %.3e" % (d["ms_per_step"], r["forward"]["launch_ms"], r["adjoint"]["launch_ms"], d["value"]))
except Exception as e:
    print("$wl adj=$adj failed:", e); print(open("$OUT/bench_${wl}_$adj.err").read()[-800:])
PY
  done
done
