"""eval_many against point-by-point evaluation at the headline size: time per point for K points per sweep"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pisa_amd import synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
ks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4, 5, 6, 8, 9, 12, 16]
wl = synthetic.Workload(n_events=n, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
rs = np.random.RandomState(1)
pts = [wl.osc_params(theta23_deg=31 + 28 * rs.rand(), dm31=1e-3 + 6e-3 * rs.rand()) for _ in range(48)]
for p in pts[:10]:
    st.eval_host(p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for p in pts:
    st.eval_host(p)
torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / len(pts)
print("serial  %.1f us/point  %.0f evals/s" % (ts * 1e6, 1 / ts))
for k in ks:
    batches = [pts[i:i + k] for i in range(0, len(pts) - k + 1, k)]
    for b in batches[:3]:
        st.eval_many(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for rep in range(3):
        for b in batches:
            st.eval_many(b)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / (3 * len(batches) * k)
    print("K=%2d  %.1f us/point  %.0f evals/s  x%.2f" % (k, t * 1e6, 1 / t, ts / t))
