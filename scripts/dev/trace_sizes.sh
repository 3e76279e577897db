# rocprofv3 kernel-trace duration of the fused kernel for a few sample sizes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 1.2e5 1.2e6 5e6 1e7; do
rm -rf gpurun_out/ts_$n
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ts_$n -o t -- python3 bench.py --events $n --legs none --no-cpu-baseline --no-batch-probe --no-drop-probe --steps 200 > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ts_$n/t_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("hist_accumulate","finalize_metric","prob3_chain","prob3_terms")):
        print("$n", r["Name"][:40], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,2), "min", round(float(r["MinNs"])/1e3,2))
PY
done
