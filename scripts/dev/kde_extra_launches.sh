# R6-7: is a C3 evaluation bound by the number of runtime calls?  N empty launches per estimator (development library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kde_extra
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
for rep in 1 2; do for n in 0 10 20 40; do
  echo -n "extra $n: "; PISA_HIP_KDE_EXTRA_LAUNCHES=$n timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 1e-12 2>&1 | grep median_ms
done; done | tee gpurun_out/kde_extra/extra.txt
