"""Set-up time of the engine (host columns -> first evaluation ready), by phase.  python scripts/dev/setup_probe.py [events] [binning]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pisa_amd import synthetic
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
binning = sys.argv[2] if len(sys.argv) > 2 else "dragon"
if os.environ.get("SETUP_PROBE_WARM_UP"):      # round 6: the runtime's first-use costs paid beside the generation of the sample
    import pisa_amd
    pisa_amd.warm_up(background=True)
wl = synthetic.Workload(n_events=n, grid=(200, 100), out_binning=binning, seed=0)
if os.environ.get("SETUP_PROBE_WARM_UP"):
    print(json.dumps(dict(warm_up_ms=round(pisa_amd.warm_up_wait(), 1))), flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = synthetic.DeviceState(wl, compact=True, time_setup=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    p = wl.osc_params()
    st.make_pseudo_data(p, seed=0)
    llh = st.eval_host(wl.osc_params(theta23_deg=44.0), "llh")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps(dict(rep=rep, ctor_ms=round(1e3 * (t1 - t0), 2), first_eval_ms=round(1e3 * (t2 - t1), 2), llh=llh,
                          **{k: round(v, 2) for k, v in st.setup_ms.items()})), flush=True)
    del st
    torch.cuda.empty_cache()
