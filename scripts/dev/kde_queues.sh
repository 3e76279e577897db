# R6-7: hardware queues (GPU_MAX_HW_QUEUES, read by the HIP runtime when it starts) x worker threads of the KDE pool
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/kde_queues
for rep in 1 2; do for q in ${QUEUES:-4 8 16}; do for w in ${WORKERS:-8 12 24}; do
  echo -n "queues $q workers $w: "; GPU_MAX_HW_QUEUES=$q PISA_KDE_WORKERS=$w timeout 300 python3 scripts/dev/c3_probe.py 1e7 14 1e-12 2>&1 | grep median_ms
done; done; done | tee -a gpurun_out/kde_queues/queues.txt
