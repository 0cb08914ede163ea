"""C2-size launches under rocprofv3 --kernel-trace: the kernel's own duration next to the per-launch time the stream
sees (bench.py's c2 record).   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2/kt -- python3 tools/c2_trace.py"""
import importlib, json, sys, torch
sys.path.insert(0, ".")
import bench
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
for (B, S, o) in ((4096, 8, 4), (64, 8, 4)):
    rec, prep, wp, tm = bench.bench_uniform(csp, dev, B, S, o, 300, 30, 2)
    print(json.dumps({"B": B, "S": S, "us_per_launch_events": round(rec["kernel_ms"] * 1e3, 2)}), flush=True)
