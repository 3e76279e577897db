cd $GRAFT_REPO_ROOT
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_WAVES=3072" "PISA_HIP_KDE_LATTICE_WAVES=4096" "PISA_HIP_KDE_LATTICE_WAVES=6144" "PISA_HIP_KDE_LATTICE_WAVES=2048" 2>&1 | grep -v prep
for w in 3072 4096 6144; do
echo "waves $w"; PISA_HIP_KDE_LATTICE_WAVES=$w timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
done
