# per-kernel time of one C3 evaluation (rocprofv3 kernel trace of scripts/dev/c3_probe.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kk -o k -- python3 scripts/dev/c3_probe.py 1e7 > gpurun_out/kk.log 2>&1
grep '"it"' gpurun_out/kk.log | cut -c1-40
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kk/k_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel sum per evaluation %.1f ms" % (tot / 4e6))
for r in rows[:14]:
    print("%-60s %5s %8.2f ms/eval %7.1f us avg" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 4e6, float(r["AverageNs"]) / 1e3))
PY
