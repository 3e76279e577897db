"""Knob sweep of the fused kernel in the 16-bit index form (kernel time by HIP events, interleaved):
workgroup count, threads per workgroup, LDS replicas."""
import os
import sys

import numpy as np
import torch

from pisa_amd import _lib, synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
wl = synthetic.Workload(n_events=n, grid=(200, 100))
lib = _lib.lib()
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
pts = [wl.osc_params(theta23_deg=38 + 0.1 * i) for i in range(60)]
configs = [dict(), dict(PISA_HIP_HIST_BLOCKS="256"), dict(PISA_HIP_HIST_BLOCKS="384"),
           dict(PISA_HIP_HIST_BLOCKS="768"), dict(PISA_HIP_HIST_BLOCKS="1024"),
           dict(PISA_HIP_HIST_BLOCKS="1024", PISA_HIP_HIST_THREADS="512"),
           dict(PISA_HIP_HIST_COPIES="2"), dict(PISA_HIP_HIST_COPIES="8"), dict(PISA_HIP_HIST_COPIES="1")]
keys = ("PISA_HIP_HIST_BLOCKS", "PISA_HIP_HIST_THREADS", "PISA_HIP_HIST_COPIES")
ref = None
for rep in range(2):
    for cfg in configs:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(cfg)
        for p in pts[:5]:
            st.eval_host(p, "llh")
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in pts]
        for (a, b), p in zip(pairs, pts):
            a.record(); b.record()
            lib.pisa_hip_profile_events(a.cuda_event, b.cuda_event)
            llh = st.eval(p, "llh")
        lib.pisa_hip_profile_events(None, None)
        torch.cuda.synchronize()
        llh = float(llh.item())
        ref = llh if ref is None else ref
        t = np.mean([a.elapsed_time(b) for a, b in pairs]) * 1e3
        print("rep %d %-60s fused %.1f us  same_llh=%s" % (rep, cfg, t, llh == ref))
