# fused kernel duration against the number of workgroups (fewer workgroups = fewer flush atomics)
for b in 256 320 384 448 512 640; do
echo "BLOCKS=$b"
PISA_HIP_HIST_BLOCKS=$b python bench.py --steps 300 --no-cpu-baseline --legs none --no-drop-probe --no-batch-probe 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' %.1f us/eval  fused %.2f us'%(1e6/d['value'],d['phase_ms']['fused_reweight_hist']*1e3))
"
done
