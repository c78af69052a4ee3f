#!/bin/bash
# Run ON THE GPU BOX [workload] [reps]: rocprofv3 kernel statistics of the full reference training step (default C4) (ODE part + K = 10 000 prior
# branch + Adam), written to gpurun_out/prof_tstep/.
R=$(pwd); OUT=$R/gpurun_out/prof_tstep; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/tstep.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
import bench, phoenix_amd
from phoenix_amd import training
wl = bench.WORKLOADS["${1:-breast}"]
dev = torch.device("cuda:0")
net, y0, t = bench.make_problem(wl, dev, 0)
print("training step ms:", bench.full_training_step_ms(wl, net, y0, t, dev, K=10000, reps=${2:-20}))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 /tmp/tstep.py > $OUT/trace.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-70s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
