#!/usr/bin/env python3
"""Copies a tools/profile_bench.sh result from gpurun_out/ (scratch) into profiles/ (tracked) and
derives the HBM traffic figure bench.py reports in roofline.traffic:

    hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
(16 B/lane) coalesced read stream (MI355X_MICROARCH.md §HBM), which is this kernel's copy-in
pattern; WRITE_SIZE is exact for 16 B/lane stores.  Collected in separate --pmc passes.
usage: tools/save_profile.py gpurun_out/prof_<tag> profiles/<name> [batch] [dispatch-name] [--latest]
`dispatch-name` is what csp_minsnap_kernel_name reports (e.g. fixed_o4_s16_f64); --latest also writes
profiles/traffic_latest.json, the file bench.py reads (together with the content hash of the kernel's sources:
a profile of older sources is reported as stale, not as the run's traffic)."""
import csv, glob, json, os, shutil, sys
latest = "--latest" in sys.argv
argv = [a for a in sys.argv if a != "--latest"]
src, dst = argv[1], argv[2]
batch = int(argv[3]) if len(argv) > 3 else 65536
dispatch = argv[4] if len(argv) > 4 else None
os.makedirs(dst, exist_ok=True)
for sub in ("kt", "fetch", "write", "sq1", "sq2", "misc"):
    for f in glob.glob(os.path.join(src, sub, "**", "*.csv"), recursive=True):
        base = os.path.basename(f).split("_", 1)[1]
        if base in ("agent_info.csv", "domain_stats.csv"):
            continue
        if base == "kernel_trace.csv":
            rows = list(csv.reader(open(f)))
            rows = rows[:1] + [r for r in rows[1:] if "minsnap" in r[7]][:8]   # a sample, not the whole trace
            csv.writer(open(os.path.join(dst, "%s_kernel_trace_sample.csv" % sub), "w")).writerows(rows)
            continue
        shutil.copy(f, os.path.join(dst, "%s_%s" % (sub, base)))
if os.path.exists(os.path.join(src, "summary.txt")):
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, "summary.txt"))

def mean_counter(sub, name):
    vals = []
    for f in glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "minsnap" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
                kern = r["Kernel_Name"]
    return (sum(vals) / len(vals) if vals else None)
fetch, write = mean_counter("fetch", "FETCH_SIZE"), mean_counter("write", "WRITE_SIZE")
avg_ns = None
for f in glob.glob(os.path.join(src, "kt", "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "minsnap" in r["Name"]:
            avg_ns, kname = float(r["AverageNs"]), r["Name"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # headline_source_sha(): bench.py reports this traffic figure only while the kernel's sources are unchanged
out = {"profile_dir": dst.rstrip("/"), "source_sha": bench.headline_source_sha(), "batch": batch, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch and write else None,
       "kernel_avg_ns_rocprof": avg_ns,
       "note": "gfx950: FETCH_SIZE doubled (wide coalesced reads); separate --pmc passes"}
out["kernel"] = dispatch
json.dump(out, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
if latest:
    json.dump(out, open(os.path.join(os.path.dirname(dst.rstrip("/")) or ".", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
