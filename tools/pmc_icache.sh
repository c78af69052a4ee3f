R=$(pwd); OUT=$R/gpurun_out/pmc_icache; rm -rf $OUT; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rhs_vjp_kernel_chain" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" | sort -u > $OUT/avail.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o run -- python3 $R/bench.py --workload breast --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/p$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "adj3" if "k1_solve_adj3" in k else ("fwd3" if "k1_solve_fwd3" in k else None)
        if not name: continue
        tot[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
for name in tot:
    print("==", name)
    for c in sorted(tot[name]):
        print("  %-28s %14.0f per launch" % (c, tot[name][c] / cnt[name][c]))
PY
cat $OUT/avail.txt | tr '\n' ' '
