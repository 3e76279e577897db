cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kde.py tests/test_gpu_kde_stage.py -x -q -m gpu 2>&1 | tail -3
for cfg in "${@}"; do
  echo "== $cfg"
  env $cfg python3 scripts/dev/kde_facts.py 1e7 12 2>&1 | grep -v "^W2026\|^E2026" | tail -1
done
python3 scripts/dev/c3_probe.py 1e7 2>&1 | grep '"it"' | cut -c1-60
