cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
for w in 3 4; do
  export PISA_HIP_EVENTS_WAVES=$w
  for cfg in "--events 1e6 --steps 20" "--events 1.25e7 --nsi --steps 6"; do
    rm -rf gpurun_out/evab
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/evab -o t -- python3 scripts/bench_events.py $cfg > gpurun_out/evab.json 2>/dev/null
    python3 - "waves $w" "$cfg" <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open("gpurun_out/evab/t_kernel_stats.csv")) if "prob3_events_kernel" in r["Name"]]
print(sys.argv[1], sys.argv[2], "| kernels:", ", ".join("%.1f us x%s" % (float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows))
PY
  done
done
