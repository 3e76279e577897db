"""Development probe: host overhead of the Pipeline path vs the bare engine."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pisa_amd.core.config_parser import parse_pipeline_config
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1.2e6
cfg = parse_pipeline_config("settings/pipeline/example_hip.cfg")
cfg[("data", "synthetic_events")]["params"].params.n_events.value = n
pipe = Pipeline(cfg, profile=True)
pipe.get_outputs()
torch.cuda.synchronize()
rs = np.random.RandomState(0)
t0 = time.perf_counter()
K = 30
for i in range(K):
    pipe.params.theta23.value = (35 + 20 * rs.rand()) * ureg.degree
    maps = pipe.get_outputs()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("Pipeline.get_outputs (osc param changed, maps to host): %.3f ms per eval at %d events" % (dt * 1e3, n))
pipe.report_profile()
dm = DistributionMaker(pipe)
data = dm.get_outputs(return_sum=True)
t0 = time.perf_counter()
for i in range(K):
    pipe.params.theta23.value = (35 + 20 * rs.rand()) * ureg.degree
    v = data.metric_total(dm.get_outputs(return_sum=True), "mod_chi2")
dt = (time.perf_counter() - t0) / K
print("DistributionMaker + metric_total: %.3f ms per eval" % (dt * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    pipe.params.theta23.value = (35 + 20 * rs.rand()) * ureg.degree
    maps = pipe.get_outputs()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
