"""Cost of the P[3][3] stores of the planned prob3 path: evaluation time with and without the
full probability arrays (the fused kernel only reads the (P_e, P_mu) gather tables)."""
import ctypes as C
import time

import numpy as np
import torch

from pisa_amd import kernels as K, synthetic

wl = synthetic.Workload(n_events=1_200_000, grid=(200, 100))
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
pts = [wl.osc_params(theta23_deg=38 + 0.03 * i) for i in range(320)]
for p in pts[:20]:
    st.eval_host(p, "llh")
a = st._lean
lib, s = a["lib"], K._stream()
for rep in range(2):
    for with_p in (True, False):
        nu, nubar = (a["nu"], a["nubar"]) if with_p else (None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pts:
            lib.pisa_hip_prob3_grid_planned(C.byref(p), a["plan"], a["energy"], a["n_e"], a["e_major"], nu, nubar,
                                            a["pepmu"], s)
        torch.cuda.synchronize()
        print("P arrays %s: %.1f us per prob3 evaluation" % ("written" if with_p else "skipped", (time.perf_counter() - t0) / len(pts) * 1e6))
