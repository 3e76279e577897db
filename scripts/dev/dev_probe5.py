"""Development probe: fused kernel time, tight loop vs inside the eval sequence,
for both resident event orders and with/without deposits (PISA_HIP_HIST_DBG=2)."""
import os
import sys

import numpy as np
import torch

from pisa_amd import _lib, synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
wl = synthetic.Workload(n_events=n, grid=(200, 100), out_binning="dragon", seed=0)
lib = _lib.lib()
rs = np.random.RandomState(0)
plist = [wl.osc_params(theta23_deg=40 + 10 * rs.rand()) for _ in range(30)]
for order in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("node", "bin")):
    st = synthetic.DeviceState(wl, sort_events=order)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    for dbg in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("0", "2")):
        os.environ["PISA_HIP_HIST_DBG"] = dbg
        for p in plist[:5]:
            st.eval(p, "llh")
        torch.cuda.synchronize()
        # tight loop
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plist]
        for a, b in pairs:
            a.record(); b.record()
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            st.accumulate()
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        tight = np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3
        for (a, b), p in zip(pairs, plist):
            a.record(); b.record()
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            st.eval(p, "llh")
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        ctx = np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3
        for (a, b), p in zip(pairs, plist):
            a.record(); b.record()
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            st.eval(p, "llh").item()
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        ctx_sync = np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3
        print(f"order={order} dbg={dbg}: tight {tight:.1f} us, in eval (async) {ctx:.1f} us, "
              f"in eval (readback each) {ctx_sync:.1f} us", flush=True)
    del st
