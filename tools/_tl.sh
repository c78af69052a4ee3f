cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/tl -o tl -- python3 /root/repo/bench.py --steps 10 --warmup 3 > /root/repo/gpurun_out/tl.log 2>&1
cd /root/repo
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f
