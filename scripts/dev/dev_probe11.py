import time, numpy as np, torch
from pisa_amd import synthetic
wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl, compact=True)
st.make_pseudo_data(wl.osc_params(), seed=0)
rs = np.random.RandomState(0)
plist = [wl.osc_params(theta23_deg=40 + 10 * rs.rand()) for _ in range(400)]
for rep in range(3):
    for spin in (0, 20000):
        st.spin_wait = spin
        for p in plist[:20]:
            st.eval_host(p)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vals = [st.eval_host(p) for p in plist]
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("spin_wait=%d: %.1f us per eval (%.0f evals/s) llh[5]=%r" % (spin, dt / len(plist) * 1e6, len(plist) / dt, vals[5]))
