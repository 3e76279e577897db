# kernels and gaps of one headline step (rocprofv3 kernel trace of bench.py --legs none): medians over the timed loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/st; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -o t -- python3 bench.py --legs none --no-cpu-baseline --no-drop-probe --no-kernel-timing "$@" > gpurun_out/st.log 2>&1
python3 - <<'PY'
import csv, glob, statistics
f = glob.glob("gpurun_out/st/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
keys = ("prob3_terms", "prob3_chain", "hist_accumulate", "finalize_metric")
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in keys)]
seq = []
for i in range(len(rows) - 4):
    n = [rows[i + j]["Kernel_Name"] for j in range(5)]
    if all(keys[j % 4] in n[j] for j in range(5)):
        s = [int(rows[i + j]["Start_Timestamp"]) for j in range(5)]; e = [int(rows[i + j]["End_Timestamp"]) for j in range(4)]
        seq.append((e[0] - s[0], s[1] - e[0], e[1] - s[1], s[2] - e[1], e[2] - s[2], s[3] - e[2], e[3] - s[3], s[4] - e[3], s[4] - s[0]))
seq = seq[len(seq) // 4:]
for k, l in enumerate(["terms", "gap", "chain", "gap", "fused", "gap", "tail", "turn-around", "step"]):
    print("%-12s median %6.2f us   mean %6.2f" % (l, statistics.median(x[k] for x in seq) / 1e3, statistics.mean(x[k] for x in seq) / 1e3))
PY
