# A/B of a PISA_HIP_HIST_DBG bit inside one GPU session (boxes differ run to run)
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s fused %.1f us prob3 %.1f us' % (d['value'], 1e3*d['phase_ms']['fused_reweight_hist'], 1e3*d['phase_ms']['prob3_grid']))"; }
for i in 1 2 3; do
  echo -n "default : "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | p
  echo -n "dbg=$1  : "; PISA_HIP_HIST_DBG=$1 python bench.py --no-cpu-baseline 2>&1 | tail -1 | p
done
