#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash tools/collect_round.sh r5
# Everything under profiles/ that DESIGN.md section 7 quotes, on ONE build of the library: the rocprofv3 passes of
# tools/collect_profiles.sh for the four workloads (kernel stats + PMC, with the library's sha256), then the bench lines
# (so that their `roofline.traffic` finds the matching PMC summary), the in-kernel segment timers and a 1-rank torchrun line.
TAG=${1:-r5}
R=$(pwd); OUT=$R/gpurun_out/round_$TAG; mkdir -p $OUT
for wl in breast yeast bcell insilico; do
  timeout 600 bash tools/collect_profiles.sh $wl $TAG > $OUT/collect_$wl.log 2>&1
done
for wl in breast yeast bcell insilico; do
  timeout 600 python bench.py --workload $wl > profiles/${TAG}_bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-cpu-baseline > profiles/${TAG}_bench_breast_torchrun_1rank.json 2> $OUT/torchrun.err
PHX_PROF=1 timeout 200 python tools/prof_segments.py breast adj > profiles/${TAG}_breast_segments_adj3.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py breast fwd > profiles/${TAG}_breast_segments_fwd3.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py breast adj 17 > profiles/${TAG}_breast_b17_segments_adj3.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py breast fwd 17 > profiles/${TAG}_breast_b17_segments_fwd3.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py yeast adj > profiles/${TAG}_yeast_segments_adj3c.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py yeast fwd > profiles/${TAG}_yeast_segments_fwd3c.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py bcell adj 128 > profiles/${TAG}_bcell_segments_adj3c_128.txt 2>&1
PHX_PROF=1 timeout 200 python tools/prof_segments.py bcell fwd > profiles/${TAG}_bcell_segments_fwd3c.txt 2>&1
timeout 400 python tools/v3c_check.py time > profiles/${TAG}_v3c_old_vs_new_time.txt 2>&1
timeout 600 python tools/v3c_check.py hbgrid > profiles/${TAG}_v3c_half_block_grid.txt 2>&1
timeout 600 python tools/v3c_check.py ntggrid > profiles/${TAG}_v3c_tiles_per_group_grid.txt 2>&1
mkdir -p $OUT/profiles; cp profiles/${TAG}_* $OUT/profiles/
ls -la $OUT/profiles | head -40
