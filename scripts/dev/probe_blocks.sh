p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused %.1f us' % (1e3*d['phase_ms']['fused_reweight_hist']))"; }
for dbg in 0 2 1; do echo -n "dbg=$dbg: "; PISA_HIP_HIST_DBG=$dbg python bench.py --no-cpu-baseline --no-drop-probe --steps 100 2>&1 | tail -1 | p; done
