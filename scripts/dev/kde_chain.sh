# the launch sequence of ONE estimator (one stream, rocprofv3 kernel trace): name, start, gap to the previous end, duration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kc; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kc -o k -- python3 scripts/dev/kde_facts.py ${1:-3e5} ${2:-2} > gpurun_out/kc.log 2>&1
python3 - <<'PY'
import csv
rows = sorted(csv.DictReader(open("gpurun_out/kc/k_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "kde_moments1" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev = t0; n = 0
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void pisa::", "").replace("pisa::", "").split("(")[0]
    if "rocprim" in name: name = "rocprim:" + ("onesweep" if "onesweep" in r["Kernel_Name"] else "other")
    print("%8.1f us  gap %6.1f  dur %6.1f  %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3, name[:60])); prev = en; n += 1
print(n, "launches,", round((prev - t0) / 1e3, 1), "us")
PY
