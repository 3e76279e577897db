"""A/B of the headline step: `pisa_hip_evaluator_eval` (one C-ABI call per evaluation, polled in C) against the
three separate calls + Python poll.  Alternating blocks in one process."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd import synthetic  # noqa: E402

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
pts = bench.param_list(wl, 520)
for p in pts[:20]:
    st.eval_host(p, "llh")
res = {True: [], False: []}
for rep in range(6):
    for mode in (True, False):
        st.one_call = mode
        for p in pts[:20]:
            st.eval_host(p, "llh")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for p in pts[20:]:
            v = st.eval_host(p, "llh")
        torch.cuda.synchronize()
        res[mode].append((time.perf_counter() - t0) / 500 * 1e6)
for mode in (True, False):
    print("one_call=%s: %s  median %.2f us" % (mode, " ".join("%.2f" % x for x in res[mode]), np.median(res[mode])))
