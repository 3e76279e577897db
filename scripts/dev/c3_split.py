"""where one C3 evaluation spends its wall time: the native batch call, the copy of the densities, numpy, the rest"""
import sys, time, json
sys.argv = [sys.argv[0], "1e7", "1"]
exec(open("/root/repo/scripts/dev/c3_probe.py").read().split("times = []")[0])
from pisa_amd import kernels as K
from pisa_amd.utils import kde_hist
T = {"native": 0.0, "batch_total": 0.0}
_nb, _kb = K.kde_lattice_batch, kde_hist.kde_histogramdd_batch
def nb(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = _nb(*a, **k); T["native"] += time.perf_counter() - t0; return r
def kb(*a, **k):
    t0 = time.perf_counter(); r = _kb(*a, **k); T["batch_total"] += time.perf_counter() - t0; return r
K.kde_lattice_batch = nb; kde_hist.kde_histogramdd_batch = kb
tot = 0.0
N = 12
for it in range(N + 1):
    if it == 1: T = {k: 0.0 for k in T}; tot = 0.0
    pipe.params.theta23.value = (42.0 + it) * ureg.degree
    torch.cuda.synchronize(); t0 = time.perf_counter()
    maps = pipe.get_outputs()
    torch.cuda.synchronize(); tot += time.perf_counter() - t0
print(json.dumps(dict(total_ms=round(tot / N * 1e3, 2), native_ms=round(T["native"] / N * 1e3, 2), batch_py_ms=round((T["batch_total"] - T["native"]) / N * 1e3, 2),
                      rest_ms=round((tot - T["batch_total"]) / N * 1e3, 2))))
