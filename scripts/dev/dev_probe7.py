"""Development probe: host-side cost of enqueuing one evaluation (Python + ctypes + HIP launch),
which sits on the critical path of a sequential fit after every LLH read-back."""
import time

import numpy as np
import torch

from pisa_amd import synthetic

wl = synthetic.Workload(n_events=10_000_000, grid=(200, 100), out_binning="dragon", seed=0)
st = synthetic.DeviceState(wl)
st.make_pseudo_data(wl.osc_params(), seed=0)
rs = np.random.RandomState(0)
plist = [wl.osc_params(theta23_deg=40 + 10 * rs.rand()) for _ in range(200)]
for p in plist[:10]:
    st.eval_host(p)


def timeit(fn, n=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(plist[i % len(plist)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


print("eval_host (sync each): host %.1f us, wall %.1f us" % timeit(lambda p: st.eval_host(p)))
print("eval (async enqueue):  host %.1f us, wall %.1f us" % timeit(lambda p: st.eval(p)))
print("compute_probs enqueue: host %.1f us, wall %.1f us" % timeit(lambda p: st.compute_probs(p)))
print("accumulate() enqueue:  host %.1f us, wall %.1f us" % timeit(lambda p: st.accumulate()))
print("tail enqueue:          host %.1f us, wall %.1f us" % timeit(lambda p: (setattr(st, "_maps_valid", False), st._tail("llh", st.metric_out))))
print("osc_params():          host %.1f us" % timeit(lambda p: wl.osc_params(theta23_deg=45.0))[0])
t0 = time.perf_counter()
for _ in range(200):
    torch.cuda.current_stream().synchronize()
print("idle stream sync: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for p in plist[:100]:
    st.eval_host(p)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
