cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_MIN_SHARES=8 PISA_HIP_KDE_LATTICE_WAVES=3072" "PISA_HIP_KDE_LATTICE_MIN_SHARES=8 PISA_HIP_KDE_LATTICE_WAVES=6144" "PISA_HIP_KDE_LATTICE_MIN_SHARES=4 PISA_HIP_KDE_LATTICE_WAVES=8192" "PISA_HIP_KDE_LATTICE_MIN_SHARES=16 PISA_HIP_KDE_LATTICE_WAVES=4096" 2>&1 | grep -v "prep\|combine"
NC=3 bash scripts/dev/kde_pmc.sh 2>&1 | grep lattice
python3 - <<PY
import json
d=json.load(open("gpurun_out/kde_pmc/kde_sq_counters.json"))["per_launch_means"]["kde_lattice_kernel"]
print({k: round(v) for k,v in d.items() if k!="derived"})
PY
rm -f gpurun_out/stamps_x.bin
PISA_HIP_KDE_LATTICE_MIN_SHARES=8 PISA_HIP_KDE_LATTICE_WAVES=6144 PISA_HIP_KDE_LATTICE_STAMPS=gpurun_out/stamps_x.bin python scripts/dev/kde_facts.py 1e7 1 > /dev/null 2>&1
python scripts/dev/kde_stamps.py gpurun_out/stamps_x.bin -2 | head -8
