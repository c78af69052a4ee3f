#!/bin/bash
# Run ON THE GPU BOX (through gpurun): GPU parity suite + short bench lines + segment timers of the solve kernels.
# usage: bash tools/gpu_check.sh [tag] [workloads...]
TAG=${1:-chk}; shift
WLS=${@:-breast}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for wl in $WLS; do
  for sg in ${STAGGERS:-default}; do
    if [ "$sg" != default ]; then export PHX_STAGGER_US=$sg; fi
    timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_${wl}_$sg.json 2> $OUT/bench_${wl}_$sg.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${wl}_$sg.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("$wl stagger=$sg ms_per_step %.3f  fwd %.3f ms  adj %.3f ms  frac %.3f  value %.3e" % (d["ms_per_step"], r["forward"]["launch_ms"], r["adjoint"]["launch_ms"], r["frac"], d["value"]))
except Exception as e:
    print("$wl stagger=$sg bench failed:", e); print(open("$OUT/bench_${wl}_$sg.err").read()[-1500:])
PY
  done
  unset PHX_STAGGER_US
  PHX_PROF=1 timeout 300 python tools/prof_segments.py $wl adj > $OUT/seg_adj_$wl.txt 2>&1
  grep -v amdgpu.ids $OUT/seg_adj_$wl.txt
done
for pw in ${PROF_WAVES:-}; do
  echo "== segment timers of wave $pw"; PHX_PROF_WAVE=$pw PHX_PROF=1 timeout 300 python tools/prof_segments.py breast adj 2>&1 | grep -v amdgpu.ids
done
