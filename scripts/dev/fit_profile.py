"""host profile of the C4 fit loop through the Pipeline boundary (point by point / stencil in one sweep)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from pisa_amd.analysis.analysis import Analysis
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.units import ureg

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
dm = DistributionMaker(bench._pipeline_cfg(n))
for name in dm.params.free.names:
    if name not in ("theta23", "deltam31"):
        dm.params.fix(name)
dm.params.theta23.value = 47.5 * ureg.degree
dm.params.deltam31.value = 2.55e-3 * ureg.eV ** 2
data = dm.get_outputs(return_sum=True).fluctuate("poisson", random_state=0)
ana = Analysis()
for batched in (False, True):
    for rep in range(2):
        dm.params.theta23.value = 42.3 * ureg.degree
        dm.params.deltam31.value = 2.457e-3 * ureg.eV ** 2
        pr = cProfile.Profile() if rep else None
        t0 = time.perf_counter()
        if pr: pr.enable()
        res = ana.fit_hypo(data, dm, "llh", reset_free=False, batched_gradient=batched)
        if pr: pr.disable()
        dt = time.perf_counter() - t0
        print("batched", batched, "evals", res.num_distributions_generated, "wall %.2f ms" % (dt * 1e3), "us/eval %.1f" % (1e6 * dt / res.num_distributions_generated))
        if pr:
            st_ = pstats.Stats(pr).sort_stats(os.environ.get("SORT", "cumulative")); st_.print_stats(28)
            if os.environ.get("CALLERS"): st_.print_callers(os.environ["CALLERS"])
