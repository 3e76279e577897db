import os, sys
import numpy as np, torch
from pisa_amd import _lib, synthetic
wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100))
lib = _lib.lib()
pts = [wl.osc_params(theta23_deg=38 + 0.1 * i) for i in range(60)]
for kw in (dict(lds_order=False), dict(lds_order=False, sort_events=False), dict(index16=False), dict(index16=False, lds_order=False)):
    st = synthetic.DeviceState(wl, compact=True, **kw)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    for rep in range(2):
        for cp in ("4", "1", "2"):
            os.environ["PISA_HIP_HIST_COPIES"] = cp
            for p in pts[:5]:
                st.eval_host(p, "llh")
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in pts]
            for (a, b), p in zip(pairs, pts):
                a.record(); b.record()
                lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
                st.eval(p, "llh")
            lib.pisa_hip_profile_events(None, None)
            torch.cuda.synchronize()
            print(kw, "copies", cp, "fused %.1f us" % (np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3))
    del st
