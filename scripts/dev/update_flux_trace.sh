cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/uft
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/uft -o t -- python3 bench.py --legs update_flux --no-cpu-baseline --no-batch-probe --no-drop-probe --steps 50 > /dev/null 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/uft/t_kernel_stats.csv")))[:7]:
    print(r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3,1))
PY
