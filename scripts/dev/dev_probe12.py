"""Development probe: LDS-bank-aware event order (engine.lds_bank_order) A/B in one session."""
import time

import numpy as np
import torch

from pisa_amd import synthetic

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
rs = np.random.RandomState(0)
plist = [wl.osc_params(theta23_deg=40 + 10 * rs.rand()) for _ in range(300)]
vals = {}
for lds in (False, True, False, True):
    st = synthetic.DeviceState(wl, compact=True, lds_order=lds)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    for p in plist[:20]:
        st.eval_host(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    v = [st.eval_host(p) for p in plist]
    dt = time.perf_counter() - t0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        st.accumulate()
    b.record()
    torch.cuda.synchronize()
    print("lds_order=%s: %.1f us per eval (%.0f evals/s), accumulate alone %.1f us, llh[3]=%r"
          % (lds, dt / len(plist) * 1e6, len(plist) / dt, a.elapsed_time(b) / 50 * 1e3, v[3]), flush=True)
    vals[lds] = v
    del st
assert vals[False] == vals[True]
print("identical LLH values")
