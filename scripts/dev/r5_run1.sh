cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -3
timeout 300 python scripts/dev/c3_probe.py 1e7 12 2>&1 | grep median
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_LG=8" "PISA_HIP_KDE_LATTICE_LG=16" "PISA_HIP_KDE_LATTICE_LG=32" 2>&1 | grep -v prep
for lg in 8 16; do
export PISA_HIP_KDE_LATTICE_LG=$lg
NC=3 bash scripts/dev/kde_pmc.sh 2>&1 | grep lattice
python3 - <<PY
import json
d=json.load(open("gpurun_out/kde_pmc/kde_sq_counters.json"))["per_launch_means"]["kde_lattice_kernel"]
print({k: round(v) for k,v in d.items() if k!="derived"})
PY
done
