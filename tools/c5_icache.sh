cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d gpurun_out/c5_ic -o c5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/c5_ic.log 2>&1 || { tail -5 gpurun_out/c5_ic.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/c5_ic/*counter_collection.csv")
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if "twist" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
