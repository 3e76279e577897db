cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/c3t; mkdir -p gpurun_out/c3t
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3t -o k -- python3 scripts/dev/c3_probe.py 1e7 8 > gpurun_out/c3t/log 2>&1
grep median gpurun_out/c3t/log
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/c3t/k_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f ms" % (tot/1e6))
for r in rows[:22]:
    print("%-60s %5s %8.2f ms %7.1f us avg" % (r["Name"].replace("void pisa::","").replace("pisa::","")[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
rm -f gpurun_out/c3t/k_kernel_trace.csv
