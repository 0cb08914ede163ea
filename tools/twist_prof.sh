# Kernel-level timeline of csp_minsnap_solve_mixed with the lane-pair sweep (CSP_MIXED_TWIST=15) and with the chunked kernels only (=0):
# one rocprofv3 kernel trace per setting over tools/twist_probe.py's five batches (orders 4 / 3 / 5 / 2 alone, then the C5 mix).
# Run ON the GPU box: gpurun -- bash tools/twist_prof.sh ; then python tools/twist_prof_read.py here.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/twp
for m in 0 15; do
export CSP_MIXED_TWIST=$m
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tw_prof_m$m -o tw -- python3 tools/twist_probe.py 65536 f32 --child $m /tmp/twp/m$m > gpurun_out/tw_prof_m$m.log 2>&1 || exit 1
done
python - <<'PY'
import csv, glob
for m in (0, 15):
    f = glob.glob(f"gpurun_out/tw_prof_m{m}/**/*kernel_stats.csv", recursive=True)
    print("mask", m, f)
    for row in csv.DictReader(open(f[0])):
        print("  %-90s calls=%s avg_us=%.1f" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
