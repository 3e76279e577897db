"""host profile of one evaluation through the Pipeline boundary (bench leg `pipeline_boundary`)"""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from pisa_amd.core.pipeline import Pipeline
from pisa_amd.core.units import ureg

n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e6
pipe = Pipeline(bench._pipeline_cfg(n))
data = sum(pipe.get_outputs()).fluctuate("poisson", random_state=0)
rs = np.random.RandomState(1)


def one():
    pipe.params.theta23.value = (31.0 + 28.0 * rs.rand()) * ureg.degree
    pipe.params.deltam31.value = (1e-3 + 6e-3 * rs.rand()) * ureg.eV ** 2
    return data.metric_total(expected_values=sum(pipe.get_outputs()), metric="llh")


for _ in range(20):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    one()
print("%.1f us per evaluation" % ((time.perf_counter() - t0) / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    one()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(40)
st.sort_stats("tottime").print_stats(25)
