p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f evals/s, fused %.1f us' % (d['value'], 1e3*d['phase_ms']['fused_reweight_hist']))"; }
for w in 1024 2048 4096 16384 65536; do echo -n "window=$w banks=32: "; PISA_LDS_WINDOW=$w python bench.py --no-cpu-baseline --no-drop-probe --steps 200 2>&1 | tail -1 | p; done
for b in 16 64; do echo -n "window=4096 banks=$b: "; PISA_LDS_BANKS=$b python bench.py --no-cpu-baseline --no-drop-probe --steps 200 2>&1 | tail -1 | p; done
for c in 1 2; do echo -n "copies=$c: "; PISA_HIP_HIST_COPIES=$c python bench.py --no-cpu-baseline --no-drop-probe --steps 200 2>&1 | tail -1 | p; done
