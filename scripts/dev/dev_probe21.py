"""Host-side timeline of one lean evaluation: time spent in each of the three C-ABI calls and in
the result poll (perf_counter, mean over 300 evaluations)."""
import ctypes as C
import time

import numpy as np
import torch

from pisa_amd import _lib, kernels as K, synthetic

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100))
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
pts = [wl.osc_params(theta23_deg=38 + 0.03 * i) for i in range(320)]
for p in pts[:20]:
    st.eval_host(p, "llh")
a = st._lean
lib, s = a["lib"], K._stream()
h = st._metric_host_np
T = np.zeros(5)
n = 0
t_prev_end = None
gaps = []
for p in pts[20:]:
    h[0] = np.nan
    t0 = time.perf_counter()
    lib.pisa_hip_prob3_grid_planned(C.byref(p), a["plan"], a["energy"], a["n_e"], a["e_major"], a["nu"],
                                    a["nubar"], a["pepmu"], s)
    t1 = time.perf_counter()
    lib.pisa_hip_reweight_hist_acc(a["cont"], a["n_cont"], a["grid"], a["nu"], a["nubar"], a["pepmu"], a["outb"],
                                   a["limbs"], a["status"], s)
    t2 = time.perf_counter()
    lib.pisa_hip_finalize_metric(a["limbs"], a["n_cont"], st.n_bins, a["hist"], a["sumw2"], 0, a["data"], a["out"],
                                 a["status"], a["mstatus"], 1, s)
    t3 = time.perf_counter()
    while True:
        v = h[0]
        if v == v:
            break
    t4 = time.perf_counter()
    T += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
    n += 1
T /= n
print("prob3 call %.1f us, fused call %.1f us, tail call %.1f us, poll %.1f us, total %.1f us" % tuple(T * 1e6))
