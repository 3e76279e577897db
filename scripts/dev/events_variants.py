import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from pisa_amd import synthetic
for n in (1e6, 1.25e7):
    wl = synthetic.Workload(n_events=int(n), grid=(10, 10), out_binning="example2d", seed=0)
    for compact in (False, True):
        st = synthetic.DeviceState(wl, osc_mode="events", compact=compact)
        st.make_pseudo_data(wl.osc_params(), seed=0)
        steps = 40 if n < 5e6 else 10
        plist = bench.param_list(wl, 3 + steps)
        for mode in ("item", "host"):
            f = (lambda p: st.eval(p).item()) if mode == "item" else (lambda p: st.eval_host(p, "llh"))
            for p in plist[:3]:
                f(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for p in plist[3:]:
                v = f(p)
            torch.cuda.synchronize()
            print("events %.3g compact %d %s: %.1f us  llh %.12g" % (n, compact, mode, (time.perf_counter() - t0) / steps * 1e6, v))
        del st
        torch.cuda.empty_cache()
