export PISA_HIP_LIB=${PISA_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/pisa_amd/libpisa_hip_dev.so}   # development build: make -C pisa_amd/csrc dev
# fused kernel launch time against the sample size (fixed cost + slope) and the debug switches
# (PISA_HIP_HIST_DBG: 2 no deposits, 4 no flush of the LDS accumulators to the global limbs)
for d in ${DBGS:-0}; do for n in ${SIZES:-1.2e6 2.5e6 5e6 1e7 2e7 4e7}; do
PISA_HIP_HIST_DBG=$d python bench.py --events $n --legs none --no-cpu-baseline --no-batch-probe --no-drop-probe --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1])
print('dbg',$d,'events',$n, 'launch_us', round(d['roofline']['avg_launch_ms']*1e3,2), 'step_us', round(d['ms_per_step']*1e3,2))"
done; done
