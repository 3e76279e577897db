cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q 2>&1 | tail -3
timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median
python scripts/dev/kde_tol_budget.py 1e7 2>&1 | grep -v Warn | grep '"tol": 1e-1[234]\|"tol": 0.0'
