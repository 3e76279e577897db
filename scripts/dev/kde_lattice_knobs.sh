# R6-8: the lattice kernel's launch shape under the FULL C3 load (eight streams), development library:
# wavefronts per launch (tuned single-stream in round 5: twice the resident set) and the floor of shares per wavefront
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kde_knobs
export PISA_HIP_LIB=$GRAFT_REPO_ROOT/pisa_amd/libpisa_hip_dev.so
for rep in 1 2; do
  for w in 3072 4096 6144 9216 12288; do
    echo -n "waves $w: "; PISA_HIP_KDE_LATTICE_WAVES=$w timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 1e-12 2>&1 | grep median_ms
  done
  for m in 4 16 32; do
    echo -n "min_shares $m: "; PISA_HIP_KDE_LATTICE_MIN_SHARES=$m timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 1e-12 2>&1 | grep median_ms
  done
  for r in 16; do
    echo -n "R $r: "; PISA_HIP_KDE_LATTICE_R=$r timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 1e-12 2>&1 | grep median_ms
  done
done | tee gpurun_out/kde_knobs/knobs.txt
