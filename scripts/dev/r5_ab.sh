cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in libpisa_hip_devA.so libpisa_hip_dev.so; do
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/$lib
echo "== $lib"
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_LG=8" 2>&1 | grep "lattice_kernel"
timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
done
done
