cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -3
for i in 1 2; do python bench.py --legs kde_c3 --no-cpu-baseline --no-drop-probe --no-batch-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kde_c3', d['legs']['kde_c3']['ms_per_step'])"; done
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_LG=8" "PISA_HIP_KDE_LATTICE_LG=16" 2>&1 | grep -v "prep\|combine"
NC=3 bash scripts/dev/kde_pmc.sh 2>&1 | grep lattice
