p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s, pipelined %.0f, fused %.1f us' % (d['value'], d['pipelined_evals_per_s'], 1e3*d['phase_ms']['fused_reweight_hist']))"; }
for b in 512 384 256 192 128; do echo -n "blocks=$b: "; PISA_HIP_HIST_BLOCKS=$b python bench.py --no-cpu-baseline --no-drop-probe 2>&1 | tail -1 | p; done
