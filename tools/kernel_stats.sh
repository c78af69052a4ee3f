#!/bin/bash
# usage (on the GPU box): bash tools/kernel_stats.sh <python script> [args...]   -> top kernels by total time
R=$(pwd); OUT=$R/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 "$R/$1" "${@:2}" > $OUT/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv,glob,re
f=glob.glob("gpurun_out/kstats/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n=re.sub(r"\(anonymous namespace\)::","",r["Name"]); n=re.sub(r"\(.*","",n)
    print("%-60s calls %4s avg %10.1f us  total %8.2f ms"%(n[:60],r["Calls"],float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6))
PY
