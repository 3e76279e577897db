"""Development probe: event-mode prob3 throughput."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pisa_amd import kernels as K
from pisa_amd import synthetic

wl = synthetic.Workload(n_events=12000, grid=(20, 10))
p = wl.osc_params()
earth = wl.layers.earth_struct()
rs = np.random.RandomState(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for n in (100000, 1000000):
    e = K.to_device(10 ** (rs.rand(n) * 3))
    cz = K.to_device(rs.rand(n) * 2 - 1)
    out = torch.empty((n, 3, 3), dtype=torch.float64, device="cuda")
    t = timeit(lambda: K.prob3_events(p, earth, 1, e, cz, out=out))
    print("prob3_events n=%d: %.3f ms -> %.2f M events/s" % (n, t, n / t / 1e3))
    # sorted by coszen (uniform layer count within a wave)
    idx = torch.argsort(cz)
    e2, cz2 = e[idx].contiguous(), cz[idx].contiguous()
    t = timeit(lambda: K.prob3_events(p, earth, 1, e2, cz2, out=out))
    print("prob3_events n=%d sorted by cz: %.3f ms -> %.2f M events/s" % (n, t, n / t / 1e3))
    nl, dens, dist = K.calc_layers(earth, cz, wl.layers.max_layers)
    t = timeit(lambda: K.propagate_array(p, 1, e, dens, dist, out=out))
    print("propagate_array n=%d (layers [N,L] in HBM): %.3f ms -> %.2f M events/s" % (n, t, n / t / 1e3))
