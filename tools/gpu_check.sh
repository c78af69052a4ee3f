#!/bin/bash
# Run ON THE GPU BOX (through gpurun): GPU parity suite + short bench lines + segment timers of the solve kernels.
# usage: bash tools/gpu_check.sh [tag]
TAG=${1:-chk}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
for wl in breast; do
  timeout 600 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  tail -c 1500 $OUT/bench_$wl.json
  PHX_PROF=1 timeout 300 python tools/prof_segments.py $wl adj > $OUT/seg_adj_$wl.txt 2>&1
  PHX_PROF=1 timeout 300 python tools/prof_segments.py $wl fwd > $OUT/seg_fwd_$wl.txt 2>&1
  cat $OUT/seg_adj_$wl.txt
done
