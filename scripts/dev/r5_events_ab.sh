cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in libpisa_hip_old.so libpisa_hip.so; do
  export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/$lib
  for cfg in "--events 1e6 --steps 20" "--events 1.25e7 --nsi --steps 6"; do
    rm -rf gpurun_out/evab
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/evab -o t -- python3 scripts/bench_events.py $cfg > gpurun_out/evab.json 2>/dev/null
    python3 - "$lib" "$cfg" <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open("gpurun_out/evab/t_kernel_stats.csv")) if "prob3_events_kernel" in r["Name"]]
line = json.loads(open("gpurun_out/evab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "| kernels:", ", ".join("%.1f us x%s" % (float(r["AverageNs"]) / 1e3, r["Calls"]) for r in rows), "| evals/s %.1f llh %r" % (line.get("value", 0), line.get("last_llh", line.get("llh"))))
PY
  done
done
unset PISA_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prob3_variants.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "event_mode or c5" 2>&1 | tail -2
