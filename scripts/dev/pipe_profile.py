"""Where the host time of Pipeline.get_outputs() + metric goes (cfg text -> maps -> LLH)."""
import cProfile, pstats, sys, io, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg
n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
cfg[("data", "synthetic_events")]["params"].params.n_events.value = n
pipe = Pipeline(cfg)
maps = pipe.get_outputs()
data = sum(maps).fluctuate("poisson", random_state=0)
th = np.linspace(40, 50, 1300)
def one(i):
    pipe.params.theta23.value = th[i] * ureg.degree
    ms = pipe.get_outputs()
    return data.metric_total(expected_values=sum(ms), metric="llh")
for i in range(20):
    one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20, 520):
    one(i)
torch.cuda.synchronize()
print("per eval: %.1f us" % ((time.perf_counter() - t0) / 500 * 1e6))
# host-side split
def t(fn, n=300):
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("set param      %.1f us" % t(lambda i: setattr(pipe.params.theta23, "value", th[600 + i] * ureg.degree)))
print("params access  %.1f us" % t(lambda i: pipe.params.theta23))
osc = pipe["prob3"]
print("_matrices      %.1f us" % t(lambda i: osc._matrices()))
def go(i):
    pipe.params.theta23.value = th[900 + i] * ureg.degree
    return pipe.get_outputs()
print("set+get_outputs (async) %.1f us" % t(go))
pr = cProfile.Profile()
pr.enable()
for i in range(520, 620):
    one(i)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
