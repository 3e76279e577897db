"""one C3-sized estimator: lattice evaluation timing (and the debug counters of a -DKDE_LATTICE_DEBUG build)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pisa_amd import kernels as K

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400000
rs = np.random.RandomState(3)
x = np.stack([np.clip(rs.rand(n) * 2 - 1 + rs.randn(n) * 0.15, -1, 1), np.log(10 ** (rs.rand(n) * 1.2 + 0.7))])
w = rs.rand(n) + 0.1
est = K.KdeEstimator(K.to_device(x), K.to_device(w), adaptive=True, alpha=0.1)
n0, n1 = 120, 80
a0 = np.linspace(-1.495, 1.495, n0)
a1 = np.linspace(np.log(5.7), np.log(55.0), n1)
args = ([a0[0], a1[0]], [a0[1] - a0[0], a1[1] - a1[0]], (n0, n1))
for _ in range(3):
    est.evaluate_lattice(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = est.evaluate_lattice(*args)
torch.cuda.synchronize()
print("lattice evaluate %.3f ms, pairs %.3e" % ((time.perf_counter() - t0) * 100, est.pairs_eval))
pts = np.array([g.ravel() for g in np.meshgrid(a0, a1, indexing="ij")])
q = K.to_device(pts)
for _ in range(2):
    ref = est(q)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ref = est(q)
torch.cuda.synchronize()
print("point evaluate   %.3f ms, pairs %.3e, max rel diff %.2e" % ((time.perf_counter() - t0) * 200, est.pairs_eval,
      float(((out - ref).abs() / ref.max()).max())))
