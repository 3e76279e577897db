"""`Container` auto-translation (events -> map, map -> events) against numpy restatements of the reference's rule
(pisa/core/container.py:933-1012; development tool, GPU box).  Every trial: a binning of 1-3 dimensions (linear, log-regular
or irregular, declared lin or log), events with coordinates on edges / outside / NaN, a scalar and a [N, 2] variable in 'sum'
or 'average' mode.  The rule: no irregular dimension -> log dimensions in ln x, fast_histogram's half-open arithmetic;
any irregular dimension -> numpy's `histogramdd` on the ORIGINAL coordinates for all dimensions (last edge included).
usage: fuzz_container.py [trials] [seed]"""
import sys

import numpy as np

sys.path.insert(0, ".")
from pisa_amd.core.binning import MultiDimBinning, OneDimBinning  # noqa: E402
from pisa_amd.core.container import Container  # noqa: E402
from pisa_amd.core.translation import find_index  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for trial in range(trials):
    ndim = int(rs.randint(1, 4))
    dims = []
    for d in range(ndim):
        kind = ["lin", "log", "irr"][rs.choice(3, p=[0.5, 0.3, 0.2])]
        n = int(rs.randint(1, 10 if ndim == 3 else 30))
        lo = rs.uniform(0.5, 10)
        if kind == "lin":
            dims.append(OneDimBinning("v%d" % d, num_bins=n, domain=[lo, lo + rs.uniform(0.5, 20)], is_lin=True))
        elif kind == "log":
            dims.append(OneDimBinning("v%d" % d, num_bins=n, domain=[lo, lo * rs.uniform(1.5, 100)], is_log=True))
        else:
            dims.append(OneDimBinning("v%d" % d, bin_edges=lo + np.concatenate([[0], np.cumsum(rs.uniform(0.1, 3, n))]),
                                      is_log=bool(rs.rand() < 0.3)))
    b = MultiDimBinning(dims)
    n_ev = int(10 ** rs.uniform(0, 4))
    c = Container("c")
    cols = {}
    for d in dims:
        e = d.edge_magnitudes
        x = rs.uniform(max(e[0] - 0.2 * (e[-1] - e[0]), 1e-3), e[-1] + 0.2 * (e[-1] - e[0]), n_ev)
        k = rs.rand(n_ev)
        x = np.where(k < 0.08, e[rs.randint(0, len(e), n_ev)], x)
        cols[d.name] = x
        c[d.name] = x
    w = rs.rand(n_ev) + 0.1
    vec = rs.rand(n_ev, 2)
    mode = ["sum", "average"][rs.randint(2)]
    c["w"], c["vec"] = w, vec
    c.translation_modes["w"] = c.translation_modes["vec"] = mode
    # ---- numpy restatement of the rule
    inside = np.ones(n_ev, dtype=bool)
    idx = []
    for d in dims:
        x, e = cols[d.name], d.edge_magnitudes
        if b.is_irregular:
            i = find_index(x, e)
            ok = (i >= 0) & (i < d.num_bins)
        else:
            y, lo, hi = (np.log(x), np.log(e[0]), np.log(e[-1])) if d.is_log else (x, e[0], e[-1])
            ok = (y >= lo) & (y < hi)
            i = np.where(ok, ((np.where(ok, y, lo) - lo) * (d.num_bins / (hi - lo))).astype(np.int64), 0)
        idx.append(np.clip(i, 0, d.num_bins - 1))
        inside &= ok
    flat = np.ravel_multi_index([i[inside] for i in idx], b.shape)
    counts = np.bincount(flat, minlength=b.size).astype(float)

    def hist_of(v):
        h = np.bincount(flat, weights=v[inside], minlength=b.size).astype(float)
        if mode == "average":
            with np.errstate(divide="ignore", invalid="ignore"):
                h = np.nan_to_num(h / counts)
        return h

    problems = []
    try:
        c.representation = b
        got = c["w"]
        want = hist_of(w)
        if not np.allclose(got, want, rtol=1e-12, atol=1e-13 * max(np.abs(want).max(), 1e-300)):
            problems.append("events -> map (%s): %d bins differ" % (mode, np.count_nonzero(~np.isclose(got, want, rtol=1e-12, atol=1e-13))))
        gv = c["vec"]
        wv = np.stack([hist_of(vec[:, 0]), hist_of(vec[:, 1])], axis=1)
        if gv.shape != wv.shape or not np.allclose(gv, wv, rtol=1e-12, atol=1e-13):
            problems.append("events -> map, vector")
        # map -> events: a binned variable looked up at the events
        vals = rs.randn(b.size)
        c["m"] = vals
        c.translation_modes["m"] = "average"
        c.representation = "events"
        back = c["m"]
        want_back = np.where(inside, vals[np.ravel_multi_index(idx, b.shape)], 0.0)
        if not np.array_equal(back, want_back):
            problems.append("map -> events: %d differ" % np.count_nonzero(back != want_back))
    except Exception as e:  # pylint: disable=broad-except
        problems.append("%s %s" % (type(e).__name__, str(e)[:200]))
    if problems:
        bad += 1
        print("MISMATCH trial %d: %s, n %d | %s" % (trial, [(d.num_bins, "log" if d.is_log else "lin", "irr" if d.is_irregular else "reg") for d in dims],
                                                     n_ev, "; ".join(problems)), flush=True)
print("fuzz_container: %d trials, %d bad" % (trials, bad))
sys.exit(1 if bad else 0)
