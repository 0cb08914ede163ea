#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel mean of every counter / kernel-trace stats.
usage: tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "minsnap"
for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-42s %-26s n=%-3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
for f in sorted(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print("%-60s calls=%s avg_ns=%s min_ns=%s max_ns=%s" % (r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
