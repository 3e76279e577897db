"""Property tests (hypothesis) of the host value types: whatever binning / map / parameter is drawn, the round trips
(JSON, pickle, repr, unit conversion) give equal objects, resampling and rebinning keep totals, slicing commutes with
the arrays, arithmetic propagates errors as first-order uncorrelated propagation says.  CPU only."""
import pickle

import numpy as np
from hypothesis import given, settings, strategies as st
from numpy import array  # noqa: F401 -- reprs are evaluated

from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.map import Map, MapSet
from pisa_amd.core.param import Param, ParamSet, Prior
from pisa_amd.core.units import ureg
from pisa_amd.utils import jsons

UNITS = ["", "GeV", "m", "deg", "s", "km / s", "eV ** 2"]
names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz_", min_size=1, max_size=8).filter(lambda s: s[0] != "_")


@st.composite
def dims(draw, name=None):
    name = name or draw(names)
    units = draw(st.sampled_from(UNITS))
    n = draw(st.integers(1, 12))
    kind = draw(st.sampled_from(["lin", "log", "irregular"]))
    lo = draw(st.floats(0.01, 50))
    span = draw(st.floats(0.1, 100))
    if kind == "lin":
        d = OneDimBinning(name, num_bins=n, domain=[lo, lo + span], is_lin=True, units=units or None)
    elif kind == "log":
        d = OneDimBinning(name, num_bins=n, domain=[lo, lo * (1 + span)], is_log=True, units=units or None)
    else:
        steps = draw(st.lists(st.floats(0.05, 5), min_size=n, max_size=n))
        d = OneDimBinning(name, bin_edges=lo + np.concatenate([[0.0], np.cumsum(steps)]), units=units or None,
                          bin_names=["b%d" % i for i in range(n)] if draw(st.booleans()) else None)
    return d


@st.composite
def binnings(draw):
    k = draw(st.integers(1, 3))
    return MultiDimBinning([draw(dims(name="d%d" % i)) for i in range(k)])


@settings(max_examples=60, deadline=None)
@given(dims())
def test_one_dim_binning_round_trips(d):
    assert eval(repr(d)) == d and pickle.loads(pickle.dumps(d)) == d
    assert OneDimBinning(**{k: v for k, v in jsons.loads(jsons.dumps(d.serializable_state)).items() if k != "is_lin"}) == d
    assert d[:] == d and d[0:len(d)] == d and sum(len(b) for b in d) == len(d)
    assert np.isclose(d.bin_widths.m.sum(), d.range.m, rtol=1e-12)
    assert np.all(d.midpoints.m > d.edge_magnitudes[:-1]) and np.all(d.weighted_centers.m < d.edge_magnitudes[1:])
    if not d.units.dimensionless:
        scaled = d.to(d.units * 1000.0) if False else d            # (same-unit conversion is the identity)
        assert scaled is d
    over = d.oversample(3)
    assert len(over) == 3 * len(d) and np.array_equal(over.edge_magnitudes[::3], d.edge_magnitudes) and d.is_compat(over)
    assert over.downsample(3) == (d if d.bin_names is None else OneDimBinning(d.name, bin_edges=d.bin_edges, is_log=d.is_log))
    for f in range(1, len(d) + 1):
        if len(d) % f == 0:
            down = d.downsample(f)
            assert down.is_compat(d) and np.isclose(down.bin_widths.m.sum(), d.bin_widths.m.sum(), rtol=1e-12)
    k = len(d) // 2
    assert np.array_equal(d[k].edge_magnitudes, d.edge_magnitudes[k:k + 2]) and d[-1] == d[len(d) - 1]


@settings(max_examples=40, deadline=None)
@given(binnings(), st.integers(0, 2 ** 31 - 1))
def test_maps_follow_their_binnings(b, seed):
    rs = np.random.RandomState(seed)
    h = rs.rand(*b.shape) * 10
    e = rs.rand(*b.shape)
    m = Map("m", h, b, error_hist=e)
    assert MultiDimBinning(**jsons.loads(jsons.dumps(b.serializable_state))) == b and eval(repr(b)) == b
    back = Map.from_json(jsons.loads(jsons.dumps(m.serializable_state)))
    assert back == m and np.allclose(back.std_devs, e, rtol=1e-15) and pickle.loads(pickle.dumps(m)) == m
    assert np.isclose(m.sum(), h.sum(), rtol=1e-12)
    for d in b.names:                                        # sums over a dimension, kept or dropped
        s = m.sum(d)
        assert (np.isclose(s, h.sum()) if b.num_dims == 1 else np.allclose(s.hist, h.sum(axis=b.index(d)), rtol=1e-12))
        kept = m.sum(d, keepdims=True)
        assert kept.shape[b.index(d)] == 1 and np.isclose(kept.hist.sum(), h.sum(), rtol=1e-12)
        assert np.isclose(np.sum(m.project(d).variances), np.sum(e ** 2), rtol=1e-12)
    factors = [next(f for f in range(n, 0, -1) if n % f == 0 and f <= 3) for n in b.shape]
    coarse = m.downsample(*factors)
    assert np.isclose(coarse.hist.sum(), h.sum(), rtol=1e-12) and np.isclose(coarse.variances.sum(), (e ** 2).sum(), rtol=1e-12)
    assert coarse.binning.is_compat(b)
    if b.num_dims > 1:
        order = list(reversed(b.names))
        r = m.reorder_dimensions(order)
        assert r.binning.names == order and np.array_equal(r.hist, np.transpose(h, list(reversed(range(b.num_dims)))))
        assert r.reorder_dimensions(b.names) == m
    idx = tuple(slice(0, max(1, n // 2)) for n in b.shape)
    assert np.array_equal(m[idx].hist, h[idx]) and m[idx].binning == b[idx]
    # first-order propagation
    c = 2.5
    assert np.allclose((m * c).std_devs, c * e) and np.allclose((m + m).std_devs, np.sqrt(2) * e)
    assert np.allclose((m / m).hist, 1.0) and np.allclose((m - m).hist, 0.0)
    assert np.allclose((m ** 2).std_devs, 2 * h * e, rtol=1e-12) and np.allclose(m.sqrt().std_devs, e / (2 * np.sqrt(h)), rtol=1e-12)
    ms = MapSet([m, Map("n", h * 2, b)])
    assert MapSet.from_json(jsons.loads(jsons.dumps(ms.serializable_state))) == ms
    assert (ms * 2)["n"] == Map("n", h * 4, b) and np.isclose(sum(ms).hist.sum(), 3 * h.sum(), rtol=1e-12)


@st.composite
def params(draw, name=None):
    name = name or draw(names)
    units = draw(st.sampled_from(UNITS))
    u = ureg.parse_units(units) if units else ureg.dimensionless
    lo = draw(st.floats(-100, 100))
    span = draw(st.floats(0.5, 50))
    value = lo + span * draw(st.floats(0.05, 0.95))
    kind = draw(st.sampled_from([None, "uniform", "gaussian", "jeffreys"]))
    if kind == "jeffreys" and lo <= 0:
        kind = "gaussian"
    prior = None if kind is None else Prior(kind, **({} if kind == "uniform" else
                                                      dict(mean=value * u, stddev=0.1 * span * u) if kind == "gaussian" else
                                                      dict(A=lo * u, B=(lo + span) * u)))
    return Param(name, value * u, prior=prior, range=[lo, lo + span] * u, is_fixed=draw(st.booleans()),
                 is_discrete=False, tex=draw(st.sampled_from([None, r"\alpha"])), help=draw(st.sampled_from(["", "h"])))


@settings(max_examples=60, deadline=None)
@given(st.lists(params(), min_size=1, max_size=6, unique_by=lambda p: p.name), st.floats(0, 1))
def test_param_sets_round_trip_and_rescale(plist, r):
    ps = ParamSet(plist)
    back = ParamSet([Param(**s) for s in jsons.loads(jsons.dumps(ps.serializable_state))])
    assert back == ps and back.values_hash == ps.values_hash and pickle.loads(pickle.dumps(ps)) == ps
    assert ps <= back and back >= ps and not ps < back
    free = ps.free
    for p in free:
        p._rescaled_value = r
        assert abs(p._rescaled_value - r) < 1e-9
        assert p.range[0] <= p.value <= p.range[1] or abs(r - 0.5) >= 0.5 - 1e-12
    ps.reset_all()
    assert ps.is_nominal and all(np.isfinite(float(np.asarray(pen))) for pen in ps.priors_penalties("llh"))
    twin = ParamSet([Param(**s) for s in jsons.loads(jsons.dumps(ps.serializable_state))])
    if len(twin):
        q = twin[0]
        mid = 0.5 * (q.range[0] + q.range[1])
        if q.value != mid:
            q.value = mid
            assert twin != ps and q not in ps
