"""Round 6: the headline loop on torch's default (null) stream against a stream of torch's pool (non-blocking): does the
legacy stream's implicit synchronisation with other blocking streams cost launch time?  One process, alternating blocks."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from pisa_amd import synthetic  # noqa: E402

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
res = {}
for name in ("default", "pool", "default", "pool"):
    ctx = torch.cuda.stream(torch.cuda.Stream()) if name == "pool" else torch.cuda.stream(torch.cuda.default_stream())
    with ctx:
        st = synthetic.DeviceState(wl, compact=True)
        st.make_pseudo_data(wl.osc_params(), seed=0)
        pts = bench.param_list(wl, 520)
        for p in pts[:20]:
            st.eval_host(p, "llh")
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for p in pts[20:]:
                v = st.eval_host(p, "llh")
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 500 * 1e6)
        res.setdefault(name, []).append((float(np.median(ts)), v))
        del st
    torch.cuda.empty_cache()
for k, v in res.items():
    print(k, ["%.2f us (llh %.10f)" % x for x in v])
