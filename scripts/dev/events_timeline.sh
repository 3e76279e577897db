cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/evt; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/evt -o t -- python3 scripts/bench_events.py --events 1e6 --steps 40 > gpurun_out/evt.log 2>&1
tail -1 gpurun_out/evt.log | cut -c1-300
python3 - <<'PY'
import csv, glob, statistics
f = glob.glob("gpurun_out/evt/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
# find steps: sequence ev0, ev1, hist, finalize
names = [r["Kernel_Name"] for r in rows]
out = []
for i in range(len(rows) - 4):
    if "prob3_events_kernel" in names[i] and "prob3_events_kernel" in names[i+1] and "hist_accumulate" in names[i+2] and "finalize_metric" in names[i+3] and "prob3_events_kernel" in names[i+4]:
        s = [int(rows[i+j]["Start_Timestamp"]) for j in range(5)]; e = [int(rows[i+j]["End_Timestamp"]) for j in range(4)]
        out.append((e[0]-s[0], s[1]-e[0], e[1]-s[1], s[2]-e[1], e[2]-s[2], s[3]-e[2], e[3]-s[3], s[4]-e[3], s[4]-s[0]))
for k, l in enumerate(["events side 0", "gap", "events side 1", "gap", "hist", "gap", "tail", "turn-around", "step"]):
    print("%-14s median %7.2f us" % (l, statistics.median(x[k] for x in out) / 1e3))
PY
