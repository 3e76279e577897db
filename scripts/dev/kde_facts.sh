# facts + per-kernel times (one stream) of the C3 estimators for several PISA_HIP_KDE_HERMITE_MIN
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=${PISA_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/pisa_amd/libpisa_hip_dev.so}   # development build: make -C pisa_amd/csrc dev
mkdir -p gpurun_out/kf
for hm in ${HM_LIST:-24 8 2 1}; do
  export PISA_HIP_KDE_HERMITE_MIN=$hm
  rm -rf gpurun_out/kf/t$hm
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kf/t$hm -o k -- python3 scripts/dev/kde_facts.py 1e7 ${NC:-12} > gpurun_out/kf/facts_$hm.log 2>&1
  echo "== HERMITE_MIN $hm"; tail -1 gpurun_out/kf/facts_$hm.log
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/kf/t$hm/k_kernel_stats.csv")))
rows = [r for r in rows if "kde" in r["Name"] or "rocprim" in r["Name"] or "rocclr" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel sum per pass over the estimators %.2f ms" % (tot / 3e6))
for r in rows[:12]:
    print("%-50s %5s %8.3f ms/pass %7.1f us avg" % (r["Name"].replace("void pisa::","").replace("pisa::","")[:50], r["Calls"], float(r["TotalDurationNs"]) / 3e6, float(r["AverageNs"]) / 1e3))
PY
done
head -30 gpurun_out/kf/facts_24.log | cut -c1-400
