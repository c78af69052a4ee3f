#!/bin/bash
# Run ON THE GPU BOX: SQ instruction-mix / wait counters of the prior-branch chain kernels (k2_*), K = 10 000 rows at C4
# (separate rocprofv3 --pmc passes, no tracing).  Output: gpurun_out/pmc_prior/summary.txt
R=$(pwd); OUT=$R/gpurun_out/pmc_prior; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o run -- python3 $R/tools/prior_branch_time.py ${1:-10000} > $OUT/p$i.log 2>&1
done
cd $R
python3 - > $OUT/summary.txt <<PY
import csv, glob, collections, re
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k2_\w+<[^>]*>|k2_\w+)", r["Kernel_Name"])
        if not m: continue
        name = m.group(1)
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
for name in sorted(tot):
    print("==", name)
    for c in sorted(tot[name]):
        print("  %-28s %14.0f per launch" % (c, tot[name][c] / cnt[name][c]))
PY
cat $OUT/summary.txt
