# PMC pass over the C5 side benchmark (run ON the GPU box): what the lane-pair sweep's waves do with their cycles
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/c5_pmc1 -o c5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/c5_pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/c5_pmc2 -o c5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/c5_pmc2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("c5_pmc1", "c5_pmc2"):
    f = glob.glob(f"gpurun_out/{d}/*counter_collection.csv")
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "twist" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s n=%d mean=%.6g" % (k, len(v), sum(v) / len(v)))
PY
