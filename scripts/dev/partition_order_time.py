"""Round 6: the partitioned resident order (binnings beyond the LDS accumulators) -- torch formulation against the native
calls, one container of the fine3d leg's size (8.3e5 events, 20 000 nodes, 4 800 bins, windows of 672 bins)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pisa_amd import engine  # noqa: E402
from pisa_amd import kernels as K  # noqa: E402

n, n_nodes, n_bins, width = 833333, 20000, 4800, 672
rs = np.random.RandomState(1)
node = rs.randint(0, n_nodes, size=n).astype(np.int32)
obin = rs.randint(0, n_bins, size=n).astype(np.int32)
obin[rs.rand(n) < 0.65] = -1
d_node, d_bin = torch.from_numpy(node).to(K.device()), torch.from_numpy(obin).to(K.device())
for name, fn in (("torch", lambda: engine.window_partition_order(d_bin, d_node, n_bins, width, n_wg=21)),
                 ("native", lambda: engine.window_partition_order_native(d_bin, d_node, n_bins, width, n_nodes, n_wg=21))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print("%-7s %.2f ms per container" % (name, 1e2 * (time.perf_counter() - t0)))
