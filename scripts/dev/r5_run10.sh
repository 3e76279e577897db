cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --legs kde_c3 --no-cpu-baseline --no-drop-probe --no-batch-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kde_c3', d['legs']['kde_c3']['ms_per_step'])"; done
timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_WAVES=6144" "PISA_HIP_KDE_LATTICE_WAVES=3072" "PISA_HIP_KDE_LATTICE_WAVES=9216 PISA_HIP_KDE_LATTICE_MIN_SHARES=4" 2>&1 | grep -v "prep\|combine"
rm -f gpurun_out/stamps_y.bin
PISA_HIP_KDE_LATTICE_STAMPS=gpurun_out/stamps_y.bin python scripts/dev/kde_facts.py 1e7 1 > /dev/null 2>&1
python scripts/dev/kde_stamps.py gpurun_out/stamps_y.bin -2 | head -6
