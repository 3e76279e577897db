"""Development probe: prob3 grid paths."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pisa_amd import synthetic

for grid in ((200, 100), (200, 200)):
    wl = synthetic.Workload(n_events=120000, grid=grid, out_binning="dragon", seed=0)
    p = wl.osc_params()
    st = synthetic.DeviceState(wl)
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

    print(grid, "planned: %.1f us" % timeit(lambda: st.compute_probs(p)))
    a = st.prob_nu.clone()
    plan, st.plan = st.plan, None
    print(grid, "direct : %.1f us" % timeit(lambda: st.compute_probs(p)))
    print(grid, "bit identical:", bool((a == st.prob_nu).all()))
