cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/evt
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/evt -o t -- python3 scripts/bench_events.py --events 1.25e7 --nsi --steps 10 > /dev/null 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/evt/t_kernel_stats.csv")))[:8]:
    print(r["Name"][:70], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1), "total_ms", round(float(r["TotalDurationNs"])/1e6,2))
PY
