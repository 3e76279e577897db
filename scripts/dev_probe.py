"""Development probe (not part of the product): time the fused kernel and the
prob3 grid stages under a few variants on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pisa_amd import synthetic

n_events = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
wl = synthetic.Workload(n_events=int(n_events), grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl)
p = wl.osc_params()
st.eval(p)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dbg in ("0", "1", "2"):
    os.environ["PISA_HIP_HIST_DBG"] = dbg
    t = timeit(lambda: st.accumulate())
    print("fused indexed dbg=%s: %.1f us  (%.2f TB/s at 40 B/event)" % (dbg, t, 40 * st.n_local / t / 1e6))
os.environ["PISA_HIP_HIST_DBG"] = "0"
print("prob3 planned: %.1f us" % timeit(lambda: st.compute_probs(p)))
st.plan = None
print("prob3 direct : %.1f us" % timeit(lambda: st.compute_probs(p)))
# plain device copy of the same volume for reference
src = torch.empty(int(40 * st.n_local // 8), dtype=torch.float64, device="cuda")
dst = torch.empty_like(src)
t = timeit(lambda: dst.copy_(src))
print("torch copy of %d MB: %.1f us (%.2f TB/s read+write)" % (src.numel() * 8 // 1e6, t, 2 * src.numel() * 8 / t / 1e6))
t = timeit(lambda: src.sum())
print("torch sum of %d MB: %.1f us (%.2f TB/s read)" % (src.numel() * 8 // 1e6, t, src.numel() * 8 / t / 1e6))
