export PISA_HIP_LIB=${PISA_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/pisa_amd/libpisa_hip_dev.so}   # development build: make -C pisa_amd/csrc dev
for t in 1024 768 512; do for b in 512 768 1024; do
echo "THREADS=$t BLOCKS=$b"
PISA_HIP_HIST_THREADS=$t PISA_HIP_HIST_BLOCKS=$b python bench.py --steps 300 --no-cpu-baseline --legs none 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(' seq %.1f us  pipelined %.1f us  fused %.1f us'%(1e6/d['value'],1e6/d['pipelined_evals_per_s'],d['phase_ms']['fused_reweight_hist']*1e3))
"
done; done
