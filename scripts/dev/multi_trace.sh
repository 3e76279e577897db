#!/bin/bash
# kernel trace of eval_many at the headline size: per-kernel durations for K points per sweep
cd /tmp && export TMPDIR=/tmp
K=${1:-5}
rm -rf /tmp/mt && mkdir -p /tmp/mt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mt -o mt -- python3 $GRAFT_REPO_ROOT/scripts/dev/multi_probe.py 1e7 $K > /tmp/mt/log.txt 2>&1
tail -3 /tmp/mt/log.txt
f=$(find /tmp/mt -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/multi_trace
cp $f $GRAFT_REPO_ROOT/gpurun_out/multi_trace/kernel_stats_K$K.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %6s avg %9.2f us  total %.1f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
