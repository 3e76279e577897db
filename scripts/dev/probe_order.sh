p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phase_ms']['fused_reweight_hist'])"; }
echo "bin full"; python bench.py --no-cpu-baseline --event-order bin 2>&1 | tail -1 | p
echo "bin nodeposit"; PISA_HIP_HIST_DBG=2 python bench.py --no-cpu-baseline --event-order bin 2>&1 | tail -1 | p
echo "bin mode3"; PISA_HIP_HIST_NO_RUNS=1 python bench.py --no-cpu-baseline --event-order bin 2>&1 | tail -1 | p
echo "node nodeposit"; PISA_HIP_HIST_DBG=2 python bench.py --no-cpu-baseline --event-order node 2>&1 | tail -1 | p
