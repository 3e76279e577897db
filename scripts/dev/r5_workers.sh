cd $GRAFT_REPO_ROOT
for w in 4 6 8 12 16; do echo "workers $w"; PISA_KDE_WORKERS=$w timeout 300 python scripts/dev/c3_probe.py 1e7 16 2>&1 | grep median; done
