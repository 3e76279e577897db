"""`Map` / `MapSet` as value types (pisa/core/map.py:187-1896, 1898-2838): shape operations (sums, projections,
rebinning, bins, splits), arithmetic with linear error propagation, comparisons, pseudo-data, set-wide
operations, JSON states.  Metrics run on the GPU and are tested there.  CPU only."""
import pickle
import re
from copy import deepcopy

import numpy as np
import pytest

from pisa_amd.core.binning import MultiDimBinning, OneDimBinning
from pisa_amd.core.map import Map, MapSet, rebin
from pisa_amd.core.units import ureg


def _binning():
    e = OneDimBinning(name="energy", tex=r"E_\nu", num_bins=10, domain=(1, 80) * ureg.GeV, is_log=True)
    cz = OneDimBinning(name="coszen", tex=r"\cos\,\theta", num_bins=5, domain=(-1, 0), is_lin=True)
    pid = OneDimBinning(name="pid", bin_edges=[0, 0.5, 1], bin_names=["cascade", "track"])
    return e, cz, pid


def test_sums_projections_rebinning():
    e, cz, _ = _binning()
    m = Map(name="x", hist=np.arange(50.0).reshape(10, 5), binning=(e, cz), error_hist=np.sqrt(np.arange(50.0).reshape(10, 5)))
    assert m.sum() == 1225.0 and m.num_entries == 1225.0 and m.size == 50 and m.item(7) == 7.0
    s = m.sum("energy", keepdims=True)
    assert np.array_equal(s.hist, [[225, 235, 245, 255, 265]]) and s.shape == (1, 5) and "energy" in s.binning
    assert np.array_equal(s.variances, s.hist)                     # variances add
    assert s.binning.energy.num_bins == 1 and np.array_equal(s.binning.energy.edge_magnitudes, e.edge_magnitudes[[0, -1]])
    s = m.sum("energy", keepdims=False)
    assert np.array_equal(s.hist, [225, 235, 245, 255, 265]) and s.shape == (5,) and "energy" not in s.binning
    assert np.array_equal(m.sum(["energy", "coszen"], keepdims=True).hist, [[1225]])
    assert np.array_equal(m.sum(1).hist, m.hist.sum(axis=1))
    assert m.project("coszen") == m.sum("energy") and m.project("energy", keepdims=True).shape == (10, 1)
    with_nan = Map(name="n", hist=[[1.0, np.nan], [2.0, 3.0]], binning=[dict(name="a", bin_edges=[0, 1, 2]), dict(name="b", bin_edges=[0, 1, 2])])
    assert with_nan.sum() == 6.0 and np.array_equal(with_nan.sum("a").hist, [3.0, 3.0])
    coarse = m.rebin(m.binning.downsample(2, 5))
    assert coarse.shape == (5, 1) and np.array_equal(coarse.hist[:, 0], m.hist.reshape(5, 10).sum(axis=1))
    assert m.rebin(m.binning.downsample(10, 5)).hist[0, 0] == 1225.0 and m.downsample(2, 1) == m.rebin(m.binning.downsample(2, 1))
    assert np.array_equal(coarse.variances, coarse.hist) and m.rebin(m.binning) == m
    swapped = m.rebin(MultiDimBinning([cz, e.downsample(5)]))       # another order of the dimensions
    assert swapped.shape == (5, 2) and np.array_equal(swapped.hist, m.hist.reshape(2, 5, 5).sum(axis=1).T)
    part = rebin(m.hist, m.binning, MultiDimBinning([e[2:6].downsample(2), cz]))      # a sub-range of the original
    assert np.array_equal(part, m.hist[2:6].reshape(2, 2, 5).sum(axis=1))
    with pytest.raises(ValueError):
        m.rebin(MultiDimBinning([e.oversample(2), cz]))                                # finer: edges not a subset
    with pytest.raises(ValueError):
        m.rebin(MultiDimBinning([e]))


def test_bins_slices_splits_and_order():
    e, cz, pid = _binning()
    h = np.arange(100.0).reshape(10, 5, 2)
    m = Map(name="x", hist=h, binning=(e, cz, pid), error_hist=np.ones((10, 5, 2)))
    assert m[0, 0, 0].shape == (1, 1, 1) and m[0, 0, 0].hist.item() == 0 and m[-1, -1, -1].hist.item() == 99
    assert np.array_equal(m[2:4, :, 1].hist, h[2:4, :, 1:2]) and m[2:4, :, 1].binning == m.binning[2:4, :, 1]
    assert np.array_equal(m["track"].hist, h[:, :, 1:2]) and np.array_equal(m.slice(pid="cascade", energy=slice(0, 3)).hist, h[0:3, :, 0:1])
    assert np.array_equal(m[2:4, :, 1].std_devs, np.ones((2, 5, 1)))
    with pytest.raises(ValueError):
        m["nonexistent"]
    assert [b.hist.item() for b in m.iterbins()][:4] == [0, 1, 2, 3] and len(list(m.itercoords())) == 100
    by_pid = m.split("pid")
    assert isinstance(by_pid, MapSet) and by_pid.names == ["cascade", "track"] and by_pid.name == "x"
    assert np.array_equal(by_pid["track"].hist, h[:, :, 1]) and by_pid["track"].binning.names == ["energy", "coszen"]
    assert np.array_equal(m.split("pid", bin="cascade").hist, h[:, :, 0]) and m.split("coszen").names[0] == "coszen_bin0"
    r = m.reorder_dimensions(["pid", "energy", "coszen"])
    assert r.shape == (2, 10, 5) and np.array_equal(r.hist, np.transpose(h, (2, 0, 1))) and r.hist.flags.c_contiguous
    assert r.reorder_dimensions(m.binning) == m
    sq = m[:, :, 0].squeeze()
    assert sq.shape == (10, 5) and sq.binning.names == ["energy", "coszen"] and np.array_equal(sq.hist, h[:, :, 0])


def test_arithmetic_and_error_propagation():
    e, cz, _ = _binning()
    b = MultiDimBinning([e, cz])
    a = Map(name="a", hist=np.full((10, 5), 4.0), binning=b, error_hist=np.full((10, 5), 0.4))
    c = Map(name="c", hist=np.full((10, 5), 2.0), binning=b, error_hist=np.full((10, 5), 0.1))
    plain = Map(name="p", hist=np.full((10, 5), 3.0), binning=b)

    def check(m, value, sigma):
        np.testing.assert_allclose(m.hist, value, rtol=1e-14)
        np.testing.assert_allclose(m.std_devs, sigma, rtol=1e-14, atol=1e-300)

    check(a + c, 6.0, np.hypot(0.4, 0.1))
    check(a - c, 2.0, np.hypot(0.4, 0.1))
    check(a * c, 8.0, 8.0 * np.hypot(0.1, 0.05))
    check(a / c, 2.0, 2.0 * np.hypot(0.1, 0.05))
    check(a + 1, 5.0, 0.4); check(1 + a, 5.0, 0.4); check(a - 1, 3.0, 0.4); check(10 - a, 6.0, 0.4)
    check(a * 3, 12.0, 1.2); check(3 * a, 12.0, 1.2); check(a / 2, 2.0, 0.2); check(8 / a, 2.0, 8 * 0.4 / 16)
    check(-a, -4.0, 0.4); check(abs(-a), 4.0, 0.4); check(a ** 2, 16.0, 2 * 4 * 0.4); check(a ** 0.5, 2.0, 0.1)
    check(a ** c, 16.0, np.hypot(2 * 4 * 0.4, 16 * np.log(4) * 0.1))
    check(a.sqrt(), 2.0, 0.1); check(a.log(), np.log(4), 0.1); check(a.log10(), np.log10(4), 0.1 / np.log(10))
    check(plain + plain, 6.0, 0.0); check(plain * a, 12.0, 1.2); check((a * 1.3).round2int(), 5.0, 0.52)
    assert (plain * 2)._var is None and sum([a, c, plain]).hist[0, 0] == 9.0
    assert a == 4.0 and a == np.full((10, 5), 4.0) and not a == 5.0 and a != c and a == deepcopy(a)
    assert pickle.loads(pickle.dumps(a)) == a
    other = deepcopy(a)
    other.set_errors(np.full((10, 5), 0.5))
    assert other != a and not a.allclose(other) and a.allclose(a + 1e-14) and a.allclose(4.0)
    # a hash stands for the contents unless a full comparison is asked for
    h1, h2 = Map(name="h", hist=np.ones((10, 5)), binning=b, hash=23), Map(name="h", hist=np.zeros((10, 5)), binning=b, hash=23)
    assert h1 == h2 and hash(h1) == 23
    h1.full_comparison = h2.full_comparison = True
    assert h1 != h2
    cmp = (a * 1.1).compare(a)
    np.testing.assert_allclose(cmp["max_abs_fract_diff"], 0.1, rtol=1e-12)
    np.testing.assert_allclose(cmp["total_abs_diff"], 20.0, rtol=1e-12)
    assert cmp["nanmatch"] and cmp["infmatch"] and cmp["ratio"].shape == (10, 5)
    with pytest.raises(ValueError):
        a.assert_compat(Map(name="z", hist=np.ones(5), binning=[cz]))


def test_pseudo_data_follow_the_seed():
    from scipy.stats import norm, poisson

    e, cz, _ = _binning()
    hist = np.linspace(0.5, 30, 50).reshape(10, 5)
    hist[3, 2] = np.nan
    m = Map(name="x", hist=hist, binning=(e, cz), error_hist=np.sqrt(hist) * 1.5)
    ok = ~np.isnan(hist)
    f = m.fluctuate("poisson", random_state=0)
    assert np.array_equal(f.hist[ok], poisson.rvs(hist[ok], random_state=np.random.RandomState(0)))
    assert np.isnan(f.hist[3, 2]) and np.isnan(f.std_devs[3, 2])
    np.testing.assert_allclose(f.std_devs[ok], np.sqrt(hist[ok]), rtol=1e-15)     # errors of the ORIGINAL expectation
    assert m.fluctuate(" Poisson ", random_state=0) == f and m.fluctuate("poisson", random_state=1) != f
    g = m.fluctuate("gauss", random_state=3)
    assert np.array_equal(g.hist[ok], norm.rvs(loc=hist[ok], scale=m.std_devs[ok], random_state=np.random.RandomState(3)))
    rs = np.random.RandomState(5)
    smeared = np.clip(norm.rvs(loc=hist[ok], scale=m.std_devs[ok], random_state=rs), 0, None)
    assert np.array_equal(m.fluctuate("gauss+poisson", random_state=5).hist[ok], poisson.rvs(smeared, random_state=rs))
    sp = m.fluctuate("scaled_poisson", random_state=7)
    scale = 1.5 ** 2
    np.testing.assert_allclose(sp.hist[ok], poisson.rvs(hist[ok] / scale, random_state=np.random.RandomState(7)) * scale, rtol=1e-14)
    np.testing.assert_allclose(sp.std_devs[ok], m.std_devs[ok], rtol=1e-15)       # the standard deviation is kept
    no_err = Map(name="x", hist=np.where(ok, hist, 0.0), binning=(e, cz))
    assert np.array_equal(no_err.fluctuate("scaled_poisson", random_state=2).hist,
                          no_err.fluctuate("poisson", random_state=2).hist)
    assert m.fluctuate(None).allclose(m) and m.fluctuate("none") is not m
    with pytest.raises(ValueError):
        m.fluctuate("binomial")
    ms = MapSet([m, (m * 2)._rebuilt(hist * 2, None, m.binning, name="y")])
    rs = np.random.RandomState(11)
    want = [mm.fluctuate("poisson", rs).hist for mm in ms]                          # ONE state through all maps
    got = ms.fluctuate("poisson", random_state=11)
    assert all(np.array_equal(a[ok], b.hist[ok]) for a, b in zip(want, got))


def test_mapset_operations():
    e, cz, _ = _binning()
    b = MultiDimBinning([e, cz])
    names = ["nue_cc", "nuebar_cc", "numu_cc", "nue_nc"]
    ms = MapSet([Map(name=n, hist=np.full((10, 5), float(i + 1)), binning=b, error_hist=np.full((10, 5), 0.1)) for i, n in enumerate(names)],
                name="set")
    assert ms.names == names and len(ms) == 4 and "numu_cc" in ms and ms.numu_cc is ms["numu_cc"] is ms[2]
    assert ms.index("nue_nc") == 3 and ms.index(ms[1]) == 1 and ms.index(-1) == 3 and ms[1:3].names == names[1:3]
    with pytest.raises(ValueError):
        ms.index("nutau_cc")
    assert ms[0, 0].names == names and ms[0, 0]["numu_cc"].hist.item() == 3.0
    # set-wide arithmetic: with numbers, and map by map with another set -- by name
    twice = ms * 2
    assert isinstance(twice, MapSet) and twice.name == "set" and twice["nue_nc"] == 8.0 and (2 * ms)["nue_nc"] == 8.0
    shuffled = MapSet([ms[n] for n in reversed(names)])
    assert [m.hist[0, 0] for m in ms + shuffled] == [2.0, 4.0, 6.0, 8.0]
    assert [m.hist[0, 0] for m in MapSet(ms.maps, collate_by_name=False) + shuffled] == [5.0] * 4      # by position
    assert [m.hist[0, 0] for m in (ms - ms)] == [0.0] * 4 and [m.hist[0, 0] for m in ms / ms] == [1.0] * 4
    assert [m.hist[0, 0] for m in (ms ** 2)] == [1.0, 4.0, 9.0, 16.0] and (-ms)["nue_cc"] == -1.0 and abs(-ms)["nue_cc"] == 1.0
    assert (10 - ms)["nue_cc"] == 9.0 and (12 / ms)["numu_cc"] == 4.0 and ms.sqrt()["nue_nc"] == 2.0
    np.testing.assert_allclose(ms.log10()["nuebar_cc"].hist, np.log10(2.0))
    assert sum(ms).hist[0, 0] == 10.0 and ms.total().name == "total" and np.isclose(ms.total().std_devs[0, 0], 0.2)
    # any Map attribute / method through the set
    assert ms.sum() == {"nue_cc": 50.0, "nuebar_cc": 100.0, "numu_cc": 150.0, "nue_nc": 200.0}
    assert ms.sum("energy").names == names and ms.sum("energy")["numu_cc"].shape == (5,)
    assert ms.project("coszen", keepdims=True)["nue_cc"].shape == (1, 5) and ms.downsample(2, 5)["nue_cc"].shape == (5, 1)
    assert ms.rebin(b.downsample(10, 5))["nue_nc"].hist.item() == 200.0 and ms.reorder_dimensions(["coszen", "energy"])[0].shape == (5, 10)
    assert ms.shape == dict.fromkeys(names, (10, 5)) and ms.num_entries["numu_cc"] == 150.0
    assert ms.apply_to_maps("allclose", ms) == dict.fromkeys(names, True)
    with pytest.raises(AttributeError):
        ms.no_such_attribute
    ms2 = deepcopy(ms)
    ms2.set_poisson_errors()
    assert np.array_equal(ms2["nue_nc"].variances, ms2["nue_nc"].hist) and ms2 != ms and ms2.allclose(ms2) and not ms2.allclose(ms)
    assert deepcopy(ms) == ms and pickle.loads(pickle.dumps(ms)) == ms
    # combinations
    cc = ms.combine_wildcard("*_cc")
    assert isinstance(cc, Map) and cc.name == "cc" and cc == 6.0 and np.isclose(cc.std_devs[0, 0], 0.1 * np.sqrt(3))
    both = ms.combine_wildcard(["nue*", "numu_cc"])
    assert isinstance(both, MapSet) and both.names == ["nue", "numu_cc"] and both["nue"] == 7.0 and both["numu_cc"] is not ms["numu_cc"]
    assert ms.combine_re(r"nue(bar)?_cc") == 3.0 and ms.combine_re(re.compile(".*")) == 10.0
    assert ms.combine_re([r"^nue_", r".*_nc$"]).names == ["nue", "nue_nc"]
    with pytest.raises(ValueError):
        ms.combine_wildcard("nutau*")
    popped = ms2.pop("nuebar_cc")
    assert popped.name == "nuebar_cc" and ms2.names == ["nue_cc", "numu_cc", "nue_nc"] and ms2.pop().name == "nue_nc"
    # hashes
    assert ms.hash is None
    ms.hash = 5
    assert ms.hashes == [5] * 4 and ms.hash == 5
    ms[0].hash = 6
    assert ms.hash == hash((6, 5, 5, 5))
    assert ms.compare(ms)["nue_cc"]["max_abs_diff"] == 0.0


def test_json_states(tmp_path):
    e, cz, pid = _binning()
    m = Map(name="x", hist=np.arange(100.0).reshape(10, 5, 2), binning=(e, cz, pid), error_hist=np.ones((10, 5, 2)), tex="X")
    m.to_json(tmp_path / "m.json")
    back = Map.from_json(tmp_path / "m.json")
    assert back == m and back.tex == "X" and back.binning == m.binning and np.array_equal(back.std_devs, m.std_devs)
    plain = Map(name="y", hist=np.ones((10, 5, 2)), binning=m.binning)
    ms = MapSet([m, plain], name="pair", tex="P")
    ms.to_json(tmp_path / "ms.json.bz2")
    back = MapSet.from_json(tmp_path / "ms.json.bz2")
    assert back == ms and back.name == "pair" and back.tex == "P" and back["y"]._var is None


def test_fit_result_state_round_trip(tmp_path):
    """`HypoFitResult` as a dictionary and as a JSON file (pisa/analysis/analysis.py:229-232, 345-371)"""
    from collections import OrderedDict

    from pisa_amd.analysis.analysis import HypoFitResult
    from pisa_amd.core.param import Param, ParamSet, Prior

    params = ParamSet(Param("theta23", 42 * ureg.deg, prior=Prior("uniform"), range=[0, 90] * ureg.deg, is_fixed=False),
                      Param("aeff_scale", 1.0, prior=Prior("gaussian", mean=1.0, stddev=0.1), range=[0, 3], is_fixed=True))
    e, cz, _ = _binning()
    total = Map(name="total", hist=np.arange(50.0).reshape(10, 5), binning=(e, cz), error_hist=np.ones((10, 5)))
    meta = OrderedDict(success=True, nit=2, nfev=7, message="converged")
    for template in (MapSet([total]), [MapSet([total]), MapSet([total * 2])]):
        fit = HypoFitResult("chi2", 1.25, params, template, [[1.5, 40.0], [1.25, 42.0]], meta, 7)
        assert fit["metric_val"] == 1.25 and fit["params"] == params and fit.params is not params      # a snapshot
        with pytest.raises(ValueError):
            fit["no_such_property"]
        fit.to_json(tmp_path / "fit.json")
        back = HypoFitResult.from_json(tmp_path / "fit.json")
        assert back.metric == "chi2" and back.metric_val == 1.25 and back.params == fit.params
        assert back.hypo_asimov_dist == fit.hypo_asimov_dist and back.fit_history == fit.fit_history
        assert back.minimizer_metadata == meta and back.num_distributions_generated == 7
        assert HypoFitResult.from_state(fit.state).params == fit.params
