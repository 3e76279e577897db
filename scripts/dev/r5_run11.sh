cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -3
for i in 1 2 3; do python bench.py --legs kde_c3 --no-cpu-baseline --no-drop-probe --no-batch-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kde_c3', d['legs']['kde_c3']['ms_per_step'])"; done
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
bash scripts/dev/kde_lat_time.sh "PISA_HIP_KDE_LATTICE_WAVES=6144" 2>&1 | tail -4
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/kl/t/k_kernel_stats.csv")):
    if "lattice_load" in r["Name"]: print("load kernel avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
