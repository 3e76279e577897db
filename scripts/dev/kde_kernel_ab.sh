# per-kernel durations of one C3 run (rocprofv3 kernel trace) for the previous and the current library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kde_kab
for lib in prev new; do
  if [ $lib = prev ]; then export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_prev.so; else unset PISA_HIP_LIB; fi
  rm -rf gpurun_out/kde_kab/$lib
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kde_kab/$lib -o k -- python3 scripts/dev/c3_probe.py 1e7 8 1e-12 > gpurun_out/kde_kab/$lib.log 2>&1
  echo "== $lib"; python3 - gpurun_out/kde_kab/$lib/k_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(x in r["Name"] for x in ("kde_lattice_kernel", "lattice_prep", "lattice_load", "lattice_combine", "hermite_coef", "h2l", "local_pilot")):
        print("  %-50s calls %4s avg %7.1f us" % (r["Name"].replace("void pisa::", "").replace("pisa::", "")[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
unset PISA_HIP_LIB
