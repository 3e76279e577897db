"""Development probe: throughput of the KDE all-pairs kernel (pisa_hip_kde_eval)."""
import numpy as np
import torch

from pisa_amd import kernels as K

rs = np.random.RandomState(0)
for d, n, m in ((2, 100_000, 9600), (2, 100_000, 100_000), (3, 100_000, 19200)):
    src = K.to_device(rs.randn(d, n))
    coef = K.to_device(np.full(n, 1.0 / n))
    s2 = K.to_device(np.ones(n))
    q = K.to_device(rs.randn(d, m))
    inv_cov = np.eye(d)
    for _ in range(2):
        K.kde_eval(src, coef, s2, q, inv_cov)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        K.kde_eval(src, coef, s2, q, inv_cov)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 5 * 1e-3
    print("D=%d N=%d M=%d: %.3f ms, %.1f G pairs/s" % (d, n, m, t * 1e3, n * m / t / 1e9))
