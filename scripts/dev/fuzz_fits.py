"""Random fits (development tool, GPU box): a maker over `example_hip.cfg` (2.4e4 synthetic events), pseudo-data at a random
truth, a random subset of one to four parameters free (theta23, deltam31, aeff_scale, delta_index, nu_nc_norm, Barr_uphor_ratio),
a random metric and minimiser (L-BFGS-B / SLSQP settings of the reference's files).  Every trial fits twice from the nominal
start -- the finite-difference stencil of every iterate in ONE sweep of the events (`metric_many`) and point by point --
and demands the same fit: equal histories (every metric value and parameter value of every evaluation, bit for bit), equal
results; on Asimov data the fit ends no worse than the truth's own value (metric plus priors' penalty).  usage: fuzz_fits.py [trials] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from pisa_amd.analysis.analysis import Analysis  # noqa: E402
from pisa_amd.core.config_parser import parse_pipeline_config  # noqa: E402
from pisa_amd.core.distribution_maker import DistributionMaker  # noqa: E402
from pisa_amd.core.pipeline import Pipeline  # noqa: E402
from pisa_amd.core.units import ureg  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
cfg[("data", "synthetic_events")]["params"].params.n_events.value = 2.4e4
maker = DistributionMaker([Pipeline(cfg)])
CAND = dict(theta23=(36, 54, "deg"), deltam31=(2.0e-3, 3.0e-3, "eV**2"), aeff_scale=(0.8, 1.3, ""), delta_index=(-0.08, 0.08, ""),
            nu_nc_norm=(0.85, 1.15, ""), Barr_uphor_ratio=(-0.8, 0.8, ""))
SETTINGS = [None, "settings/minimizer/l-bfgs-b_ftol2e-5_gtol1e-5_eps1e-4_maxiter200.json",
            "settings/minimizer/slsqp_ftol1e-6_eps1e-4_maxiter1000.json"]
ana = Analysis()
bad = 0
for trial in range(trials):
    free = list(rs.choice(list(CAND), size=int(rs.randint(1, 5)), replace=False))
    for p in maker.params:
        p.is_fixed = p.name not in free
    maker.reset_all()
    truth = {}
    for n in free:
        lo, hi, u = CAND[n]
        truth[n] = rs.uniform(lo, hi) * (ureg.parse_units(u) if u else ureg.dimensionless)
        maker.params[n].value = truth[n]
    total = maker.get_outputs(return_sum=True)
    data = type(total)([total[0]._new(total[0].hist.copy(), None, name="total")])
    metric = ["chi2", "mod_chi2", "llh"][rs.randint(3)]
    settings = SETTINGS[rs.randint(len(SETTINGS))]
    tag = "trial %d: free %s, %s, %s" % (trial, free, metric, settings.split("/")[-1] if settings else "default")
    try:
        at_truth = ana._total_metric(data, maker.get_outputs(return_sum=True), maker, metric)    # (the priors' penalty there)
        maker.reset_free()
        start = ana._total_metric(data, maker.get_outputs(return_sum=True), maker, metric)
        a = ana.fit_hypo(data, maker, metric, minimizer_settings=settings, reset_free=True, batched_gradient=True)
        va = [p.value.m for p in maker.params.free]
        b = ana.fit_hypo(data, maker, metric, minimizer_settings=settings, reset_free=True, batched_gradient=False)
        vb = [p.value.m for p in maker.params.free]
        ha, hb = np.array(a.fit_history, dtype=float), np.array(b.fit_history, dtype=float)
        problems = []
        if ha.shape != hb.shape or not np.array_equal(ha, hb):
            k = 0
            if ha.shape == hb.shape:
                k = int(np.argmax(np.any(ha != hb, axis=1)))
            problems.append("histories differ (%s vs %s evaluations, first at %d)" % (len(ha), len(hb), k))
        if a.metric_val != b.metric_val or va != vb:
            problems.append("results differ: %r vs %r" % (a.metric_val, b.metric_val))
        sign = -1 if metric == "llh" else 1
        # the fit may end below the truth's value (the priors pull), not above it by more than the minimiser's tolerance
        slack = 2e-3 * abs(start - at_truth) + 1e-3
        # (with theta23 or deltam31 free a LOCAL minimiser may end in another octant / another oscillation maximum: that is what
        # the octant, range and grid strategies are for -- both modes end there alike)
        if "theta23" not in free and "deltam31" not in free and sign * a.metric_val > sign * at_truth + slack:
            problems.append("fit ends at %.5g, the truth has %.5g (start %.5g)" % (a.metric_val, at_truth, start))
        if problems:
            bad += 1
            print("MISMATCH", tag, "|", "; ".join(problems), flush=True)
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR", tag, type(e).__name__, str(e)[:300], flush=True)
print("fuzz_fits: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
