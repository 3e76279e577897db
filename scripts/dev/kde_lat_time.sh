# lattice kernel average duration (one stream, rocprof trace) for environment variants
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kl
for cfg in "${@}"; do
  rm -rf gpurun_out/kl/t
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kl/t -o k -- python3 scripts/dev/kde_facts.py 1e7 12 > gpurun_out/kl/log 2>&1
  echo "== $cfg"; python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/kl/t/k_kernel_stats.csv")):
    if "kde_lattice_kernel" in r["Name"] or "lattice_prep" in r["Name"] or "lattice_combine" in r["Name"]:
        print("   %-40s calls %s avg %.1f us" % (r["Name"].replace("void pisa::", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
