# round 6: the batched lattice launches (libpisa_hip.so) against the previous library (pisa_amd/libpisa_hip_prev.so, built
# from the commit before) on ONE box: C3 probe at 1e-12 and at the default 1e-14, alternating, maps digests compared
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kde_ab
for rep in 1 2; do for tol in 1e-12 1e-14; do for lib in prev new; do
  if [ $lib = prev ]; then export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_prev.so; else unset PISA_HIP_LIB; fi
  echo -n "$lib tol $tol: "; timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 $tol 2>&1 | grep median_ms
done; done; done | tee gpurun_out/kde_ab/ab.txt
unset PISA_HIP_LIB
