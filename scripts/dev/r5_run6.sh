cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -2
timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_WAVES=6144" "PISA_HIP_KDE_LATTICE_WAVES=9216" "PISA_HIP_KDE_LATTICE_WAVES=12288 PISA_HIP_KDE_LATTICE_MIN_SHARES=4" 2>&1 | grep -v "prep"
for w in 9216 12288; do
echo "waves $w"; PISA_HIP_KDE_LATTICE_WAVES=$w PISA_HIP_KDE_LATTICE_MIN_SHARES=4 timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
done
