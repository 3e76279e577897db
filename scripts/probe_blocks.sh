p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s, fused %.1f us, frac %.3f' % (d['value'], 1e3*d['phase_ms']['fused_reweight_hist'], d['roofline']['frac']))"; }
for n in 2.5e6 5e6 1e7 2e7 4e7; do echo -n "events=$n: "; python bench.py --no-cpu-baseline --no-drop-probe --steps 100 --events $n 2>&1 | tail -1 | p; done
