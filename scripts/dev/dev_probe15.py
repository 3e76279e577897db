"""A/B of event-column layouts of the fused kernel, interleaved, kernel time by HIP events:
0 compact (MODE 5, 24 B/event), 1 16-bit indices + quad-blocked flux (MODE 7, 20 B/event).
(A pair-blocked variant of MODE 5 measured no different from MODE 5 and was dropped.)"""
import os
import sys
import time

import numpy as np
import torch

from pisa_amd import _lib, synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
wl = synthetic.Workload(n_events=n, grid=(200, 100))
lib = _lib.lib()
sts, res = {}, {}
data = None
for m in ("0", "1"):
    st = synthetic.DeviceState(wl, compact=True, index16=m == "1")
    if data is None:
        st.make_pseudo_data(wl.osc_params(), seed=0)
        data = st.data.cpu().numpy()
    else:
        st.set_data(data)
    sts[m] = st
pts = [wl.osc_params(theta23_deg=38 + 0.1 * i) for i in range(100)]
for rep in range(3):
    for m, st in sts.items():
        for p in pts[:10]:
            st.eval_host(p, "llh")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pts:
            llh = st.eval_host(p, "llh")
        dt = (time.perf_counter() - t0) / len(pts)
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in pts]
        for (a, b), p in zip(pairs, pts):
            a.record(); b.record()
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            st.eval(p, "llh")
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        k = np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3
        print("rep %d layout %s: eval %.1f us, fused kernel %.1f us, llh %r" % (rep, m, dt * 1e6, k, llh))
        if rep == 0:
            st.accumulate(pts[-1]); st.finalize()
            res[m] = (llh,) + st.maps()
for k in ("1",):
    print(k, "same llh", res["0"][0] == res[k][0], "same maps", np.array_equal(res["0"][1], res[k][1]),
          np.array_equal(res["0"][2], res[k][2]))
