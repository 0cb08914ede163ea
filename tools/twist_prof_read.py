import csv
for m in (15,):
    rows = list(csv.DictReader(open(f"gpurun_out/tw_prof_m{m}/tw_kernel_trace.csv")))
    rows = [r for r in rows if "csp::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    calls = []
    for r in rows:
        if "count_kernel" in r["Kernel_Name"]:
            calls.append([])
        calls[-1].append(r)
    out = []
    for case in range(5):
        c = calls[case * 14 + 8]
        t0 = int(c[0]["Start_Timestamp"])
        out.append("%.1f(%.1f)" % ((int(c[-1]["End_Timestamp"]) - t0) / 1e3, (int(c[-1]["End_Timestamp"]) - int(c[-1]["Start_Timestamp"])) / 1e3))
    print("o4,o3,o5,o2,mixed total(twist kernel):", " ".join(out))
