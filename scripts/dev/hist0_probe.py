"""generic histogram entry point (pisa_hip_histogram_regular, MODE 0 kernel): time and effective
bandwidth for 1e7 events, 1-3 output dimensions, with and without weights"""
import sys
import time

import numpy as np
import torch

from pisa_amd import _lib, kernels as K, synthetic

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
D = synthetic.DRAGON
rs = np.random.RandomState(0)
cols = [K.to_device(rs.uniform(D["mins"][k] - 0.1, D["maxs"][k] + 0.1, n)) for k in range(3)]
w = K.to_device(rs.rand(n))
for nd in (3, 2, 1):
    b = _lib.make_binning(D["mins"][:nd], D["maxs"][:nd], D["nbins"][:nd])
    for weights in (w, None):
        for _ in range(3):
            out = K.histogram_regular(cols[:nd], weights, b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = K.histogram_regular(cols[:nd], weights, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        nbytes = 8 * n * (nd + (weights is not None))
        print("dims %d weights %d: %.1f us  %.2f TB/s  sum %.6g" % (nd, weights is not None, dt * 1e6, nbytes / dt / 1e12,
                                                                    float(out.sum())))
