cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py -x -q 2>&1 | tail -2
timeout 300 python scripts/dev/c3_probe.py 1e7 12 2>&1 | grep median
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_LG=8" "PISA_HIP_KDE_LATTICE_LG=16" 2>&1 | grep -v prep
for lg in 8 16; do
rm -f gpurun_out/stamps_$lg.bin
PISA_HIP_KDE_LATTICE_LG=$lg PISA_HIP_KDE_LATTICE_STAMPS=gpurun_out/stamps_$lg.bin python scripts/dev/kde_facts.py 1e7 1 > /dev/null 2>&1
python scripts/dev/kde_stamps.py gpurun_out/stamps_$lg.bin -2
done
