#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_profiles.sh <workload> <tag>
# Three separate rocprofv3 passes of the same bench command (kernel trace + stats; PMC FETCH_SIZE; PMC WRITE_SIZE),
# then tools/summarize_profiles.py writes profiles/<tag>_<workload>_{kernel_stats.csv,pmc_hbm.json}.
set -u
WL=${1:-breast}; TAG=${2:-r1}
R=$(pwd)
OUT=$R/gpurun_out/prof_${TAG}_${WL}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 "$R/bench.py" --workload "$WL" --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-extras > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o run -- python3 "$R/bench.py" --workload "$WL" --steps 10 --warmup 2 --prewarm-seconds 0.05 --no-cpu-baseline --no-extras > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o run -- python3 "$R/bench.py" --workload "$WL" --steps 10 --warmup 2 --prewarm-seconds 0.05 --no-cpu-baseline --no-extras > "$OUT/write.log" 2>&1
cd "$R"
python3 tools/summarize_profiles.py "$OUT" "$WL" "$TAG"
