"""wall-clock shares of the C4 fit through the Pipeline boundary (stencil in one sweep), by wrapping the layers with timers"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pisa_amd.analysis.analysis import Analysis
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.units import ureg
from pisa_amd.core import fastplan, distribution_maker
from pisa_amd import engine
from pisa_amd.stages.osc import prob3

T = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t0
            T[label + "#"] = T.get(label + "#", 0) + 1
    setattr(obj, name, g)

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
dm = DistributionMaker(bench._pipeline_cfg(n))
for name in dm.params.free.names:
    if name not in ("theta23", "deltam31"):
        dm.params.fix(name)
dm.params.theta23.value = 47.5 * ureg.degree
dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=0)
ana = Analysis()
wrap(distribution_maker.DistributionMaker, "metric_many", "dm.metric_many")
wrap(fastplan.FastPlan, "metric_many", "plan.metric_many")
wrap(engine.HotPathEngine, "eval_many", "engine.eval_many")
wrap(engine.HotPathEngine, "_many_tail", "engine._many_tail (incl. GPU wait)")
wrap(engine.HotPathEngine, "_many_sweep", "engine._many_sweep (launches)")
wrap(distribution_maker.DistributionMaker, "_set_rescaled_free_params", "set_rescaled")
wrap(prob3.prob3, "_matrices", "prob3._matrices")
wrap(Analysis, "_minimizer_callable_with_gradient", "analysis.callable_with_gradient")
for rep in range(3):
    T.clear()
    dm.params.theta23.value = 42.3 * ureg.degree
    dm.params.deltam31.value = 2.457e-3 * ureg.eV ** 2
    t0 = time.perf_counter()
    res = ana.fit_hypo(data, dm, "llh", reset_free=False, batched_gradient=True)
    dt = time.perf_counter() - t0
print("fit wall %.2f ms, %d evaluations" % (dt * 1e3, res.num_distributions_generated))
for k in sorted(T):
    if not k.endswith("#"):
        print("  %-42s %7.2f ms  %4d calls  %6.1f us/call" % (k, T[k] * 1e3, T[k + "#"], 1e6 * T[k] / T[k + "#"]))
