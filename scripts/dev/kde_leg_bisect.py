import sys, gc, time
sys.path.insert(0, "/root/repo")
import torch, bench
from pisa_amd import synthetic
which = sys.argv[1]
if which in ("state", "state_del"):
    wl = synthetic.Workload(n_events=int(1e7), grid=(200, 100), out_binning="dragon", seed=0)
    st = synthetic.DeviceState(wl, rank=0, world_size=1, indexed=True, sort_events=True, compact=True, index16=True)
    st.make_pseudo_data(wl.osc_params(), seed=0)
    for p in bench.param_list(wl, 30):
        st.eval_host(p, "llh")
    if which == "state_del":
        del st, wl
        gc.collect(); torch.cuda.empty_cache()
if which == "hostmem":
    import numpy as np
    junk = [np.random.rand(int(1e7)) for _ in range(12)]
if which == "events":
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.zeros(10, device="cuda"); b.record(); torch.cuda.synchronize(); print(a.elapsed_time(b))
r = bench.leg_kde(torch, 1e7, 16)
print(which, "kde leg ms:", round(r["ms_per_step"], 2))
