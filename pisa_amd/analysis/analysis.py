"""Fit-loop driver: the reference's minimiser callable around this build's
template evaluation (counterpart of `Analysis._minimizer_callable`,
pisa/analysis/analysis.py:2493-2670, and the scipy strategy `_fit_scipy`,
:1561-1780; minimizer settings as in `settings/minimizer/*.json`).

    metric_val = data.metric_total(hypo.get_outputs(return_sum=True), metric)
                 + hypo.params.priors_penalty(metric)
    return sign * metric_val          (sign = -1 for llh-type metrics)

Free parameters are exposed to scipy rescaled to [0, 1] (param.py:358-400).
The metric itself is evaluated on the GPU (`pisa_hip_metric`).
"""
from collections import OrderedDict

from copy import deepcopy

import numpy as np

from pisa_amd.core.param import CHI2_METRICS, LLH_METRICS

__all__ = ["Analysis", "Counter", "HypoFitResult"]


class Counter:
    def __init__(self, i=0):
        self.count = i

    def __iadd__(self, inc):
        self.count += inc
        return self


class HypoFitResult:
    def __init__(self, metric, metric_val, params, hypo_asimov_dist, fit_history, minimizer_result,
                 num_distributions_generated):
        self.metric, self.metric_val = metric, metric_val
        # a snapshot (analysis.py:356-372 deep-copies too): later fits move the maker's own objects
        self.params = deepcopy(params)
        for m in (hypo_asimov_dist if hypo_asimov_dist is not None else ()):
            m.hist   # device-backed maps are brought home: the engine's buffers are reused by the next fit
        self.hypo_asimov_dist = hypo_asimov_dist
        self.fit_history = fit_history
        self.minimizer_metadata = minimizer_result
        self.num_distributions_generated = num_distributions_generated

    detailed_metric_info = None

    @staticmethod
    def get_detailed_metric_info(data_dist, hypo_maker, hypo_asimov_dist, params, metric, other_metrics=None,
                                 detector_name=None, include_maps_binned=False):
        """per metric (the fit's and `other_metrics`): its value map by map, the priors' penalties parameter by
        parameter and, if asked for, the per-bin values as maps (analysis.py:373-459; generalized_poisson_llh and
        weighted_chi2 are not built)"""
        from pisa_amd.core.map import Map, MapSet

        others = [] if other_metrics is None else ([other_metrics] if isinstance(other_metrics, str) else list(other_metrics))
        first = metric[0] if isinstance(metric, (list, tuple)) else metric
        info = OrderedDict()
        if detector_name is not None:
            info["detector_name"] = detector_name
        for m in sorted(set([first] + others)):
            d = OrderedDict()
            d["maps"] = data_dist.metric_per_map(expected_values=hypo_asimov_dist, metric=m)
            if include_maps_binned:
                hists = data_dist.metric_per_map(expected_values=hypo_asimov_dist, metric="binned_" + m)
                d["maps_binned"] = MapSet([Map(name=a.name, hist=np.reshape(hists[k], a.shape), binning=a.binning)
                                           for a, k in zip(hypo_asimov_dist, hists)])
            d["priors"] = params.priors_penalties(metric=first)
            info[m] = d
        return info

    def add_detailed_metric_info(self, data_dist, hypo_maker=None, other_metrics=None, include_maps_binned=False):
        """fills `detailed_metric_info` from the data and this result's template and parameters"""
        self.detailed_metric_info = self.get_detailed_metric_info(
            data_dist, hypo_maker, self.hypo_asimov_dist, self.params, self.metric, other_metrics=other_metrics,
            detector_name=getattr(hypo_maker, "detector_name", None) if hypo_maker is not None else None,
            include_maps_binned=include_maps_binned)
        return self.detailed_metric_info

    def __repr__(self):
        vals = ", ".join("%s=%s" % (p.name, p.value) for p in self.params.free)
        return "HypoFitResult(%s=%.8g; %s)" % (self.metric, self.metric_val, vals)

    # -- dictionary access and files (analysis.py:229-232, 345-371) --------------------------------
    _state_attrs = ("metric", "metric_val", "params", "hypo_asimov_dist", "num_distributions_generated",
                    "minimizer_metadata", "fit_history")

    def __getitem__(self, key):
        if key in self._state_attrs:
            return getattr(self, key)
        raise ValueError("Unknown property %s" % key)

    @property
    def serializable_state(self):
        """the result as plain containers: the parameters' states, the template's maps, the minimiser's report and
        the history (first column the metric, then the free parameters in order)"""
        hypo = self.hypo_asimov_dist
        if isinstance(hypo, list):
            hypo = [h.serializable_state for h in hypo]
        elif hypo is not None:
            hypo = hypo.serializable_state
        return OrderedDict([("metric", self.metric), ("metric_val", float(self.metric_val)),
                            ("params", self.params.serializable_state), ("hypo_asimov_dist", hypo),
                            ("num_distributions_generated", self.num_distributions_generated),
                            ("minimizer_metadata", None if self.minimizer_metadata is None
                             else OrderedDict(self.minimizer_metadata)),
                            ("fit_history", self.fit_history)])

    state = serializable_state

    def to_json(self, filename, **kwargs):
        from pisa_amd.utils import jsons

        jsons.to_json(self.serializable_state, filename, **kwargs)

    @classmethod
    def from_state(cls, state):
        from pisa_amd.core.map import MapSet
        from pisa_amd.core.param import Param, ParamSet

        assert set(state) == set(cls._state_attrs), "ill-formed state dict"
        hypo = state["hypo_asimov_dist"]
        if isinstance(hypo, list):
            hypo = [MapSet.from_json(h) for h in hypo]
        elif hypo is not None:
            hypo = MapSet.from_json(hypo)
        params = ParamSet([Param(**p) for p in state["params"]])
        return cls(state["metric"], state["metric_val"], params, hypo, state["fit_history"],
                   state["minimizer_metadata"], state["num_distributions_generated"])

    @classmethod
    def from_json(cls, filename):
        from pisa_amd.utils import jsons

        return cls.from_state(jsons.from_json(filename))


def load_minimizer_settings(settings):
    """`settings`: a dict {method, options}, or the reference's minimizer-settings format
    (`settings/minimizer/*.json`: {"method": {"value": ..., "desc": ...}, "options": {"value":
    {...}, "desc": {...}}}, analysis.py:2560-2575), or the resource path of such a file."""
    if isinstance(settings, str):
        import json

        from pisa_amd.utils.resources import find_resource

        with open(find_resource(settings)) as fh:
            settings = json.load(fh)
    out = {}
    for key in ("method", "options"):
        if key in settings:
            val = settings[key]
            out[key] = val["value"] if isinstance(val, dict) and "value" in val else val
    return out


BOUNDED_METHODS = ("l-bfgs-b", "slsqp", "tnc", "trust-constr", "cobyla", "cobyqa")


def local_minimizer_arguments(ms, n_free, hypo_maker):
    """bounds, constraints and options of the local scipy fit from minimiser settings `ms` -- ONE place for `fit_hypo` and
    for the fit with an external penalty (they had drifted apart: bounds for cobyla / cobyqa, `constraints` left among
    the options).  -> (method in lower case, options without "constraints", bounds or None, constraints or ())"""
    method = ms["method"].lower()
    options = dict(ms.get("options", {}))
    # "constraints" among the options (analysis.py:1687-1696): SLSQP, COBYLA, trust-constr
    constrs = options.pop("constraints", None) or []
    if constrs:
        constrs = scipy_constraints_to_callables([dict(c) for c in constrs], hypo_maker)
    bounds = [(0.0, 1.0)] * n_free
    return method, options, (bounds if method in BOUNDED_METHODS else None), (constrs or ())


def scipy_constraints_to_callables(constr_dicts, hypo_maker):
    """constraints given as functions (or strings that evaluate to functions) of the ParamSet become functions of the
    rescaled free-parameter vector, in place (configure_scipy_minimization.py:274-302)"""
    from collections.abc import Mapping, Sequence
    from functools import partial

    def constr_func(x, constr_func_params):
        hypo_maker._set_rescaled_free_params(x)  # pylint: disable=protected-access
        return constr_func_params(hypo_maker.params)

    assert isinstance(constr_dicts, Sequence)
    for cd in constr_dicts:
        assert isinstance(cd, Mapping) and "fun" in cd
        constr = cd["fun"]
        fn = constr if callable(constr) else eval(constr)  # pylint: disable=eval-used
        if not callable(fn):
            raise TypeError("Evaluated object not a callable, but %s." % type(fn))
        cd["fun"] = partial(constr_func, constr_func_params=fn)
    return constr_dicts


class Analysis:
    def __init__(self):
        self._nit = 0
        self.pprint = False
        self.blindness = False

    @staticmethod
    def _sign(metric):
        """-1 for likelihoods (maximised), +1 for chi-squares; a list (one metric per detector) must be of one
        kind (analysis.py:926-935)"""
        metrics = [metric] if isinstance(metric, str) else list(metric)
        if metrics and all(m in LLH_METRICS for m in metrics):
            return -1
        if metrics and all(m in CHI2_METRICS for m in metrics):
            return +1
        raise ValueError("Defined metrics are not compatible")

    @staticmethod
    def _metrics(metric, n):
        """one metric per detector: a name for all of them, or a list of `n` names (analysis.py:913-924)"""
        if isinstance(metric, str):
            return [metric] * n
        metric = list(metric)
        if len(metric) == 1:
            return metric * n
        assert len(metric) == n, "one metric per detector"
        return metric

    def _total_metric(self, data_dist, hypo, hypo_maker, metric):
        """metric of the template(s) against the data plus the priors' penalty (analysis.py:2588-2630): summed over
        the detectors of a `Detectors` (a metric each, the penalty by the first), over the selections of a
        variable binning, or the one total"""
        if type(hypo_maker).__name__ == "Detectors":
            ms = self._metrics(metric, len(hypo_maker.distribution_makers))
            val = 0.0
            for d, h, m in zip(data_dist, hypo, ms):
                val += d.metric_total(expected_values=h, metric=m)
            return val + hypo_maker.params.priors_penalty(metric=ms[0])
        m = metric if isinstance(metric, str) else metric[0]
        if isinstance(hypo, list):
            val = 0.0
            for d, h in zip(data_dist, hypo):
                val += d.metric_total(expected_values=h, metric=m)
            return val + hypo_maker.params.priors_penalty(metric=m)
        return data_dist.metric_total(expected_values=hypo, metric=m) + hypo_maker.params.priors_penalty(metric=m)

    def _minimizer_callable(self, scaled_param_vals, hypo_maker, data_dist, metric, counter,
                            fit_history, flip_x0=None, external_priors_penalty=None):
        sign = self._sign(metric)
        x = np.asarray(scaled_param_vals, dtype=np.float64)
        if flip_x0 is not None:
            x = np.where(flip_x0, 1 - x, x)
        hypo_maker._set_rescaled_free_params(np.clip(x, 0.0, 1.0))  # pylint: disable=protected-access
        hypo = hypo_maker.get_outputs(return_sum=True)
        metric_val = self._total_metric(data_dist, hypo, hypo_maker, metric)
        counter += 1
        if fit_history is not None:
            fit_history.append([metric_val] + [p.value.m for p in hypo_maker.params.free])
        if external_priors_penalty is not None:
            metric_val += external_priors_penalty(hypo_maker=hypo_maker, metric=metric)
        if self.pprint:
            print("%6d %12.5e | %s" % (counter.count, metric_val,
                                       " ".join("%12.5e" % p.value.m for p in hypo_maker.params.free)))
        return sign * metric_val

    @staticmethod
    def _forward_stencil(x0, eps, lb, ub):
        """The points scipy's '2-point' finite differences evaluate around `x0` for an absolute step
        `eps` inside the bounds [lb, ub] -- what L-BFGS-B and SLSQP do when no Jacobian is given
        (scipy.optimize._numdiff.approx_derivative: step eps, replaced by sqrt(machine eps) * max(1, |x|)
        where x + eps == x; the sign of a step flips where the forward point would leave the bounds, and
        where neither direction has room the step shrinks to the wider side).  Returns (points, dx):
        points[0] = x0, points[1 + i] = x0 with coordinate i moved, dx[i] the step actually taken
        (recomputed as the representable difference, as scipy does).

        Follows scipy's private `_numdiff` rules as of scipy 1.9 ... 1.15 (this image: 1.15.3);
        `tests/test_host_logic.py::test_forward_stencil_is_scipys_two_point_scheme` compares it with the installed
        scipy's own `approx_derivative` on random points, bounds and steps, and
        `tests/test_gpu_pipeline.py` pins `fit_hypo(batched_gradient=True)` to the unbatched fit (x, fun, nfev, history)
        for L-BFGS-B and SLSQP: a scipy release that changes the scheme fails those tests instead of silently
        changing fit trajectories (`batched_gradient=False` is always available)."""
        # plain Python floats (IEEE doubles, the same operations as scipy's array code): for the handful
        # of free parameters of a fit the array version costs more than the arithmetic
        x0 = [float(v) for v in x0]
        n = len(x0)
        root_eps = float(np.sqrt(np.finfo(np.float64).eps))
        pts, dx = [np.array(x0, dtype=np.float64)], np.empty(n)
        for i in range(n):
            xi, lo, hi = x0[i], float(lb[i]), float(ub[i])
            h = float(eps)
            if (xi + h) - xi == 0:
                h = root_eps * (1.0 if xi >= 0 else -1.0) * max(1.0, abs(xi))
            lower, upper = xi - lo, hi - xi
            x = xi + h
            violated = x < lo or x > hi
            fitting = abs(h) <= max(lower, upper)
            if violated and fitting:
                h = -h
            elif not fitting:
                h = upper if upper >= lower else -lower
            x1 = list(x0)
            x1[i] = xi + h
            dx[i] = x1[i] - xi
            pts.append(np.array(x1, dtype=np.float64))
        return pts, dx

    def _minimizer_callable_with_gradient(self, scaled_param_vals, hypo_maker, data_dist, metric, counter,
                                          fit_history, eps, bounds):
        """`_minimizer_callable` together with the forward-difference gradient the minimiser would
        otherwise take point by point: the n + 1 points of the stencil are INDEPENDENT template
        evaluations and go through `hypo_maker.metric_many` -- one sweep of the events where the
        pipeline allows it.  Same points, same metric values, same differences and quotients as scipy's
        own finite differences, so the fit follows the same trajectory; the fit history lists the
        points in the order scipy would have asked for them."""
        sign = self._sign(metric)
        x0 = np.asarray(scaled_param_vals, dtype=np.float64)
        lb = np.array([b[0] for b in bounds], dtype=np.float64)
        ub = np.array([b[1] for b in bounds], dtype=np.float64)
        pts, dx = self._forward_stencil(x0, eps, lb, ub)
        free = hypo_maker.params.free
        at = {}
        vals = hypo_maker.metric_many(pts, data_dist, metric,
                                      on_point=lambda i: at.setdefault(i, [p.value.m for p in free]))
        for i, (x, v) in enumerate(zip(pts, vals)):
            counter += 1
            if fit_history is not None:
                fit_history.append([v] + at[i])
            if self.pprint:
                print("%6d %12.5e | %s" % (counter.count, v, " ".join("%12.5e" % xi for xi in x)))
        f = np.array([sign * v for v in vals])
        return f[0], (f[1:] - f[0]) / dx

    def _gradient_only_callable(self, scaled_param_vals, hypo_maker, data_dist, metric, counter, fit_history,
                                eps, bounds, last):
        """The forward-difference gradient alone, for a method whose line search asks for function
        values without gradients (SLSQP): the value at `x` is the one the minimiser has just been given
        by `_minimizer_callable` (`last` = [x, sign * metric]), only the n moved points are evaluated --
        together.  Evaluates the central point as well if it is not the last one seen."""
        x0 = np.asarray(scaled_param_vals, dtype=np.float64)
        if last[0] is None or not np.array_equal(last[0], x0):
            f0, g = self._minimizer_callable_with_gradient(x0, hypo_maker, data_dist, metric, counter,
                                                           fit_history, eps, bounds)
            last[0], last[1] = x0.copy(), f0
            return g
        sign = self._sign(metric)
        lb = np.array([b[0] for b in bounds], dtype=np.float64)
        ub = np.array([b[1] for b in bounds], dtype=np.float64)
        pts, dx = self._forward_stencil(x0, eps, lb, ub)
        free = hypo_maker.params.free
        at = {}
        vals = hypo_maker.metric_many(pts[1:], data_dist, metric,
                                      on_point=lambda i: at.setdefault(i, [p.value.m for p in free]))
        for i, v in enumerate(vals):
            counter += 1
            if fit_history is not None:
                fit_history.append([v] + at[i])
        return (np.array([sign * v for v in vals]) - last[1]) / dx

    def fit_hypo(self, data_dist, hypo_maker, metric, minimizer_settings=None, reset_free=True,
                 batched_gradient=True):
        """scipy.optimize.minimize over the free params (L-BFGS-B by default, as
        settings/minimizer/l-bfgs-b_ftol2e-5_gtol1e-5_eps1e-4_maxiter200.json).
        `batched_gradient` (L-BFGS-B / SLSQP without a user Jacobian): the finite-difference stencil of
        every iterate is evaluated in one call (`_minimizer_callable_with_gradient`); the fit is the
        same fit, point for point."""
        from scipy import optimize

        if reset_free:
            hypo_maker.reset_free()
        ms = dict(method="L-BFGS-B", options=dict(ftol=2e-5, gtol=1e-5, eps=1e-4, maxiter=200))
        if minimizer_settings:
            ms.update(load_minimizer_settings(minimizer_settings))
        free = hypo_maker.params.free
        if len(free) == 0:
            hypo = hypo_maker.get_outputs(return_sum=True)
            val = self._total_metric(data_dist, hypo, hypo_maker, metric)
            return HypoFitResult(metric, val, hypo_maker.params, hypo, [], None, 1)
        x0 = np.array(free._rescaled_values, dtype=np.float64)
        bounds = [(0.0, 1.0)] * len(x0)
        counter, history = Counter(), []
        # a perfect match of data and template at the starting point (pseudo-data generated at the
        # nominal values): no fit (analysis.py:1746-1786; comparisons.ALLCLOSE_KW)
        hypo = hypo_maker.get_outputs(return_sum=True)

        def maps_of(x):     # the maps of a MapSet, of a list of MapSets (detectors / selections), or the one map
            if isinstance(x, list):
                return [m for ms in x for m in maps_of(ms)]
            return list(x) if hasattr(x, "maps") else [x]
        data_maps, hypo_maps = maps_of(data_dist), maps_of(hypo)
        if len(data_maps) == len(hypo_maps) and all(
                d.hist.shape == h.hist.shape
                and np.allclose(d.hist, h.hist, rtol=1e-12, atol=np.finfo(np.float64).eps, equal_nan=True)
                for d, h in zip(data_maps, hypo_maps)):
            val = self._total_metric(data_dist, hypo, hypo_maker, metric)
            meta = OrderedDict(success=True, nit=0, nfev=0, message="Initial hypo matches data, no need for fit")
            return HypoFitResult(metric, val, hypo_maker.params, hypo, None, meta, 0)
        method, options, method_bounds, constrs = local_minimizer_arguments(ms, len(x0), hypo_maker)
        if (not constrs and batched_gradient and method in ("l-bfgs-b", "slsqp") and hasattr(hypo_maker, "metric_many")
                and "jac" not in ms and "finite_diff_rel_step" not in options):
            # the step the method would use itself: `eps` (L-BFGS-B default 1e-8, SLSQP default sqrt(eps))
            eps = options.get("eps", 1e-8 if method == "l-bfgs-b" else np.sqrt(np.finfo(np.float64).eps))
            if method == "l-bfgs-b":
                # every point the method visits needs value AND gradient: n + 1 points per call
                res = optimize.minimize(
                    fun=self._minimizer_callable_with_gradient, x0=x0, jac=True,
                    args=(hypo_maker, data_dist, metric, counter, history, eps, bounds),
                    bounds=bounds, method=ms["method"], options=options)
            else:
                # SLSQP's line search asks for values alone: single points as before, the n moved
                # points of a gradient together
                last = [None, None]

                def fun(x):
                    f = self._minimizer_callable(x, hypo_maker, data_dist, metric, counter, history)
                    last[0], last[1] = np.array(x, dtype=np.float64), f
                    return f

                def jac(x):
                    return self._gradient_only_callable(x, hypo_maker, data_dist, metric, counter, history,
                                                        eps, bounds, last)

                res = optimize.minimize(fun=fun, x0=x0, jac=jac, bounds=bounds, method=ms["method"],
                                        options=options)
        else:
            res = optimize.minimize(
                fun=self._minimizer_callable, x0=x0, args=(hypo_maker, data_dist, metric, counter, history),
                bounds=method_bounds, constraints=constrs, method=ms["method"], options=options)
        hypo_maker._set_rescaled_free_params(np.clip(res.x, 0.0, 1.0))  # pylint: disable=protected-access
        hypo = hypo_maker.get_outputs(return_sum=True)
        val = self._sign(metric) * res.fun
        meta = OrderedDict(success=bool(res.success), nit=int(getattr(res, "nit", -1)),
                           nfev=int(res.nfev), message=str(res.message))
        return HypoFitResult(metric, val, hypo_maker.params, hypo, history, meta, counter.count)

    # -- a simple global scheme: both octants of a mixing angle (analysis.py:974-1088) ------------
    @staticmethod
    def get_separate_octant_params(hypo_maker, angle_name, inflection_point, tolerance=None):
        """the angle parameter as it is, confined to the octant of its nominal value, and confined to
        the other octant with the value mirrored at `inflection_point` (manipulate_params.py:44-123)"""
        from pisa_amd.core.units import ureg

        angle = hypo_maker.params[angle_name]
        angle.reset()
        angle_orig = angle            # the maker's own object: it is put back after the octant fits
        octants = ((angle.range[0], inflection_point), (inflection_point, angle.range[1]))
        if tolerance is None:
            tolerance = 0.1 * ureg.degree
        dist = (angle.value - inflection_point).m_as("rad")
        if abs(dist) < tolerance.m_as("rad"):
            angle.value = inflection_point + (-1.0 if dist < 0.0 else 1.0) * tolerance
        case1, case2 = deepcopy(angle), deepcopy(angle)
        first = 0 if case1.value.m_as("rad") < inflection_point.m_as("rad") else 1
        case1.range = octants[first]
        case1.nominal_value = case1.value
        mirrored = (2 * inflection_point - case2.value).to(case2.value.units)
        # range before value: a Param validates its value against its range
        case2.range = octants[1 - first]
        case2.value = mirrored
        case2.nominal_value = case2.value
        return angle_orig, case1, case2

    def fit_octants(self, data_dist, hypo_maker, metric, angle="theta23", inflection_point=None,
                    tolerance=None, minimizer_settings=None, reset_free=True):
        """Fit with `angle` confined to either octant and keep the better fit (`_fit_octants`,
        analysis.py:974-1088: the local minimiser does not cross the octant degeneracy of theta23 by
        itself).  The maker ends with its original parameter OBJECT at the best-fit values."""
        from pisa_amd.core.units import ureg

        if angle not in hypo_maker.params.free.names:
            return self.fit_hypo(data_dist, hypo_maker, metric, minimizer_settings, reset_free=reset_free)
        if inflection_point is None:
            inflection_point = 45.0 * ureg.degree
        if reset_free:
            hypo_maker.reset_free()
        start = [(p.name, p.value) for p in hypo_maker.params.free]
        orig, case1, case2 = self.get_separate_octant_params(hypo_maker, angle, inflection_point, tolerance)
        hypo_maker.update_params(case1)
        best = self.fit_hypo(data_dist, hypo_maker, metric, minimizer_settings, reset_free=False)
        for name, value in start:
            if name != angle:
                hypo_maker.params[name].value = value
        hypo_maker.update_params(case2)
        other = self.fit_hypo(data_dist, hypo_maker, metric, minimizer_settings, reset_free=False)
        sign = self._sign(metric)   # +1: smaller is better
        if sign * other.metric_val < sign * best.metric_val:
            best, other = other, best
        best.alternate_fit = other
        for res in (best, other):   # the results carry the original range, not an octant's
            res.params[angle].range = deepcopy(orig.range)
        hypo_maker.update_params(orig)
        for p in best.params.free:
            hypo_maker.params[p.name].value = p.value
        return best

    # -- nested fit strategies (analysis.py:854-1560) -------------------------------------------------
    # A strategy is {method, method_kwargs, local_fit_kwargs}; `local_fit_kwargs` is again such a dict (or a
    # list of them) that the strategy runs as its inner fit(s).  `scipy` is the local fit; `iminuit` and
    # `nlopt` need their packages.

    @staticmethod
    def _update_values(hypo_maker, params, update_nominal_values=False, update_range=False, update_is_fixed=False):
        """values (optionally ranges, nominal values, fixed flags) of `params` onto the maker's OWN parameter
        objects of the same names (manipulate_params.py:126-158; `_detector` form :161-190)"""
        from pisa_amd.core.param import Param, ParamSet

        if isinstance(params, Param):
            params = [params]
        if type(hypo_maker).__name__ == "Detectors":
            for d in hypo_maker:
                ps = ParamSet([deepcopy(p) for p in params])
                for name in list(ps.names):
                    tail = "_%s" % d.detector_name
                    if name.endswith(tail):
                        plain = name[: -len(tail)]
                        if plain in ps.names:
                            ps.remove(plain)
                        ps[name].name = plain
                        ps._reindex()  # pylint: disable=protected-access
                Analysis._update_values(d, list(ps), update_nominal_values, update_range, update_is_fixed)
            hypo_maker.init_params()
            return
        for p in params:
            for pipeline in hypo_maker:
                if p.name not in pipeline.params.names:
                    continue
                own = pipeline.params[p.name]
                if update_range:
                    own.range = p.range
                own.value = p.value
                if update_nominal_values:
                    own.nominal_value = p.nominal_value
                if update_is_fixed:
                    own.is_fixed = p.is_fixed

    def _better(self, new, old, metric):
        sign = self._sign(metric)
        return sign * new < sign * old

    def _inner(self, data_dist, hypo_maker, metric, external_priors_penalty, spec, store_fit_history,
               include_metric_maps):
        return self.fit_recursively(data_dist, hypo_maker, metric, external_priors_penalty, spec["method"],
                                    spec.get("method_kwargs"), spec.get("local_fit_kwargs"),
                                    store_fit_history=store_fit_history, include_metric_maps=include_metric_maps)

    def fit_recursively(self, data_dist, hypo_maker, metric, external_priors_penalty, method, method_kwargs=None,
                        local_fit_kwargs=None, store_fit_history=False, include_metric_maps=False):
        """Global search strategies around local fits, nested to any depth: `method` one of scipy, octants,
        best_of, condition, grid_scan, constrained, ranges, staged (iminuit, nlopt: their packages).  Returns
        the `HypoFitResult` of the best fit; the maker is left at its values."""
        n = len(hypo_maker.distribution_makers) if type(hypo_maker).__name__ == "Detectors" else 1
        metrics = [metric] if isinstance(metric, str) else list(metric)
        if type(hypo_maker).__name__ == "Detectors":
            if len(metrics) == 1:
                metrics = metrics * n
            elif len(metrics) != n:
                raise IndexError("Number of defined metrics does not match with number of detectors.")
        else:
            assert len(metrics) == 1, "Only one metric allowed for DistributionMaker"
        if method in ("fit_octants", "fit_ranges"):
            method = method.split("_")[1]
        fit = getattr(self, "_fit_%s" % method, None)
        if fit is None:
            raise ValueError("unknown fit method '%s'" % method)
        return fit(data_dist, hypo_maker, metrics if n > 1 else metrics[0], external_priors_penalty,
                   method_kwargs or {}, local_fit_kwargs, store_fit_history, include_metric_maps)

    def _fit_scipy(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                   store_fit_history, include_metric_maps):
        """a local scipy minimiser; `method_kwargs` are minimiser settings in the reference's form
        ({"method": {"value": ...}, "options": {"value": {...}}}) or a settings file"""
        if method_kwargs.get("global_method") is not None:
            res = self._fit_scipy_global(data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs,
                                         local_fit_kwargs)
            if not store_fit_history:
                res.fit_history = None
            return res
        settings = {k: v for k, v in method_kwargs.items() if k in ("method", "options")} or None
        if external_priors_penalty is None:
            res = self.fit_hypo(data_dist, hypo_maker, metric, minimizer_settings=settings, reset_free=False)
        else:
            res = self._fit_with_penalty(data_dist, hypo_maker, metric, settings, external_priors_penalty)
        if not store_fit_history:
            res.fit_history = None
        return res

    GLOBAL_SCIPY_METHODS = ("differential_evolution", "basinhopping", "dual_annealing", "shgo")

    def _fit_scipy_global(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs):
        """scipy's global optimisers over the [0, 1]-rescaled free parameters (analysis.py:1594-1680, 1811-1893):
        `method_kwargs = {"global_method": name, "options": {...}}`, the options handed to scipy as they are;
        basinhopping (with the reference's bounded random displacement, manipulate_params.py:18-41), dual_annealing
        and shgo polish with the local minimiser of `local_fit_kwargs` (settings in the reference's form) if given.
        Every evaluation is the minimiser callable of the local fits.  Constraints are not built."""
        from scipy import optimize
        from sklearn.utils import check_random_state

        name = method_kwargs["global_method"]
        if name not in self.GLOBAL_SCIPY_METHODS:
            raise ValueError("Unsupported global fit method %s" % name)
        opt = dict(method_kwargs.get("options") or {})
        if opt.pop("constraints", None):
            raise NotImplementedError("constraints with global scipy methods are not part of this build")
        uses_local = name in ("basinhopping", "dual_annealing", "shgo")
        local = None
        if uses_local and local_fit_kwargs is not None:
            ms = load_minimizer_settings({k: v for k, v in local_fit_kwargs.items() if k in ("method", "options")})
            if dict(ms.get("options", {})).get("constraints"):
                raise NotImplementedError("constraints with global scipy methods are not part of this build")
            local = dict(method=ms["method"], options=dict(ms.get("options", {})))
        free = hypo_maker.params.free
        if len(free) == 0:
            hypo = hypo_maker.get_outputs(return_sum=True)
            return HypoFitResult(metric, self._total_metric(data_dist, hypo, hypo_maker, metric), hypo_maker.params, hypo,
                                 [], None, 1)
        x0 = np.array(free._rescaled_values, dtype=np.float64)
        bounds = [(0.0, 1.0)] * len(x0)
        counter, history = Counter(), []
        sign = self._sign(metric)

        def fun(x, *unused):
            return self._minimizer_callable(np.clip(x, 0.0, 1.0), hypo_maker, data_dist, metric, counter, history, None,
                                            external_priors_penalty)

        self._nit = 0

        def count(*unused_args, **unused_kwargs):
            self._nit += 1

        if name == "differential_evolution":
            res = optimize.differential_evolution(func=fun, bounds=bounds, callback=count, **opt)
        elif name == "basinhopping":
            rng = check_random_state(opt.get("seed"))
            stepsize = opt.get("stepsize", 0.5)
            lo, hi = np.array(bounds).T

            def take_step(x):
                x += rng.uniform(-stepsize, stepsize, np.shape(x))
                return np.clip(x, lo, hi)

            minimizer_kwargs = dict(local, bounds=bounds) if local is not None else {"bounds": bounds}
            res = optimize.basinhopping(func=fun, x0=x0, take_step=take_step, callback=count,
                                        minimizer_kwargs=minimizer_kwargs, **opt)
        elif name == "dual_annealing":
            res = optimize.dual_annealing(func=fun, bounds=bounds, x0=x0, callback=count,
                                          minimizer_kwargs=dict(local, bounds=bounds) if local is not None else {}, **opt)
        else:
            res = optimize.shgo(func=fun, bounds=bounds, callback=count,
                                minimizer_kwargs=dict(local, bounds=bounds) if local is not None else {}, **opt)
        success = bool(getattr(res, "success", True))
        if name == "basinhopping":          # its OptimizeResult reports through the last local result
            success = bool(getattr(getattr(res, "lowest_optimization_result", res), "success", True))
        hypo_maker._set_rescaled_free_params(np.clip(res.x, 0.0, 1.0))  # pylint: disable=protected-access
        hypo = hypo_maker.get_outputs(return_sum=True)
        meta = OrderedDict(success=success, nit=int(getattr(res, "nit", self._nit)), nfev=int(getattr(res, "nfev", counter.count)),
                           message=str(getattr(res, "message", "")), global_method=name)
        return HypoFitResult(metric, sign * float(res.fun), hypo_maker.params, hypo, history, meta, counter.count)

    def _fit_iminuit(self, *args, **kwargs):
        raise ImportError("the 'iminuit' strategy needs the iminuit package, which is not installed; use 'scipy'")

    def _fit_nlopt(self, *args, **kwargs):
        raise ImportError("the 'nlopt' strategy needs the nlopt package, which is not installed; use 'scipy'")

    def _fit_with_penalty(self, data_dist, hypo_maker, metric, settings, external_priors_penalty):
        """the local fit with a user penalty added to every evaluation (`_minimizer_callable`'s
        `external_priors_penalty`, analysis.py:2640-2646): point by point"""
        from scipy import optimize

        ms = dict(method="L-BFGS-B", options=dict(ftol=2e-5, gtol=1e-5, eps=1e-4, maxiter=200))
        if settings:
            ms.update(load_minimizer_settings(settings))
        x0 = np.array(hypo_maker.params.free._rescaled_values, dtype=np.float64)  # pylint: disable=protected-access
        counter, history = Counter(), []
        _, options, method_bounds, constrs = local_minimizer_arguments(ms, len(x0), hypo_maker)
        res = optimize.minimize(
            fun=self._minimizer_callable, x0=x0,
            args=(hypo_maker, data_dist, metric, counter, history, None, external_priors_penalty),
            bounds=method_bounds, constraints=constrs, method=ms["method"], options=options)
        hypo_maker._set_rescaled_free_params(np.clip(res.x, 0.0, 1.0))  # pylint: disable=protected-access
        hypo = hypo_maker.get_outputs(return_sum=True)
        meta = OrderedDict(success=bool(res.success), nit=int(getattr(res, "nit", -1)), nfev=int(res.nfev),
                           message=str(res.message))
        return HypoFitResult(metric, self._sign(metric) * res.fun, hypo_maker.params, hypo, history, meta, counter.count)

    def _fit_octants(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                     store_fit_history, include_metric_maps):
        """both octants of a mixing angle, the better inner fit kept (analysis.py:974-1092)"""
        angle = method_kwargs["angle"]
        args = (data_dist, hypo_maker, metric, external_priors_penalty, local_fit_kwargs, store_fit_history,
                include_metric_maps)
        if angle not in hypo_maker.params.free.names:
            return self._inner(*args)
        reset_free = method_kwargs.get("reset_free", True)
        start = None if reset_free else deepcopy(hypo_maker.params)
        orig, case1, case2 = self.get_separate_octant_params(hypo_maker, angle, method_kwargs["inflection_point"],
                                                             method_kwargs.get("tolerance"))
        hypo_maker.update_params(case1)
        best = self._inner(*args)
        if reset_free:
            hypo_maker.reset_free()
        else:
            for p in start:
                if p.name != angle:
                    hypo_maker.params[p.name].value = p.value
        hypo_maker.update_params(case2)
        other = self._inner(*args)
        for res in (best, other):
            res.params[angle].range = deepcopy(orig.range)
        if self._better(other.metric_val, best.metric_val, metric):
            best, other = other, best
        best.alternate_fit = other
        hypo_maker.update_params(orig)
        self._update_values(hypo_maker, list(best.params.free), update_range=True)
        return best

    def _fit_best_of(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                     store_fit_history, include_metric_maps):
        """several configured fits (`local_fit_kwargs` is a list), the best one returned (analysis.py:1094-1132)"""
        results = []
        for spec in local_fit_kwargs:
            if method_kwargs.get("reset_free", True):
                hypo_maker.reset_free()
            results.append(self._inner(data_dist, hypo_maker, metric, external_priors_penalty, spec,
                                       store_fit_history, include_metric_maps))
        sign = self._sign(metric)
        best = results[int(np.argmin([sign * r.metric_val for r in results]))]
        self._update_values(hypo_maker, list(best.params.free))
        return best

    def _fit_condition(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                       store_fit_history, include_metric_maps):
        """the first of two strategies if `condition_func(hypo_maker)` holds, else the second (analysis.py:1134-1170)"""
        assert "condition_func" in method_kwargs and len(local_fit_kwargs) == 2
        cond = method_kwargs["condition_func"]
        if isinstance(cond, str):
            cond = eval(cond)  # pylint: disable=eval-used
        if not callable(cond):
            raise ValueError("Condition function is neither a callable nor a string that can be evaluated to a callable.")
        spec = local_fit_kwargs[0] if cond(hypo_maker) else local_fit_kwargs[1]
        return self._inner(data_dist, hypo_maker, metric, external_priors_penalty, spec, store_fit_history,
                           include_metric_maps)

    def _fit_grid_scan(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                       store_fit_history, include_metric_maps):
        """the inner fit started from (or, `fix_grid_params`, held at) every point of a grid of parameter values;
        optionally a `refined_fit` from the best point with everything free again (analysis.py:1172-1290)"""
        grid = method_kwargs["grid"]
        names = list(grid)
        reset_free = method_kwargs.get("reset_free", True)
        fix = method_kwargs.get("fix_grid_params", False)
        mesh = np.meshgrid(*[np.atleast_1d(grid[n].m) for n in names])
        if reset_free:
            hypo_maker.reset_free()
        originally_free = list(hypo_maker.params.free.names)
        results = []
        for idx in np.ndindex(mesh[0].shape):
            if reset_free:
                hypo_maker.reset_free()
            for n, m in zip(names, mesh):
                moved = deepcopy(hypo_maker.params[n])
                moved.value = m[idx] * grid[n].u
                if fix:
                    moved.is_fixed = True
                self._update_values(hypo_maker, moved, update_is_fixed=True)
            results.append(self._inner(data_dist, hypo_maker, metric, external_priors_penalty, local_fit_kwargs,
                                       store_fit_history, include_metric_maps))
        for n in originally_free:
            self._set_fixed(hypo_maker, n, False)
        sign = self._sign(metric)
        best = results[int(np.argmin([sign * r.metric_val for r in results]))]
        best.grid_metric_vals = np.array([r.metric_val for r in results]).reshape(mesh[0].shape)
        self._update_values(hypo_maker, [p for p in best.params if p.name in originally_free])
        refined = method_kwargs.get("refined_fit")
        if refined is not None:
            spec = dict(refined)
            spec["method_kwargs"] = dict(spec.get("method_kwargs") or {}, reset_free=False) \
                if spec["method"] != "scipy" else spec.get("method_kwargs")
            best = self._inner(data_dist, hypo_maker, metric, external_priors_penalty, spec, store_fit_history,
                               include_metric_maps)
        return best

    @staticmethod
    def _set_fixed(hypo_maker, name, flag):
        makers = hypo_maker if type(hypo_maker).__name__ == "Detectors" else [hypo_maker]
        for maker in makers:
            for pipeline in maker:
                if name in pipeline.params.names:
                    pipeline.params[name].is_fixed = flag
        if type(hypo_maker).__name__ == "Detectors":
            hypo_maker.init_params()

    def _fit_constrained(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                         store_fit_history, include_metric_maps):
        """the inner fit under `ineq_func(params) > 0`: its violation times a penalty is added to the metric and
        the penalty doubled until the fit ends inside (analysis.py:1292-1394)"""
        assert "ineq_func" in method_kwargs and "necessary_free_params" in method_kwargs
        args = (store_fit_history, include_metric_maps)
        if not set(method_kwargs["necessary_free_params"]).issubset(hypo_maker.params.free.names):
            return self._inner(data_dist, hypo_maker, metric, external_priors_penalty, local_fit_kwargs, *args)
        ineq = method_kwargs["ineq_func"]
        if isinstance(ineq, str):
            ineq = eval(ineq)  # pylint: disable=eval-used
        if not callable(ineq):
            raise ValueError("Inequality function is neither a callable nor a string that can be evaluated to a callable.")

        def violation(params):
            v = ineq(params)
            return 0.0 if v > 0.0 else -v

        penalty = method_kwargs.get("minimum_penalty", 1000.0)
        tol = method_kwargs.get("constraint_tol", 1e-4)
        sign = self._sign(metric)
        if method_kwargs.get("reset_free", False):
            hypo_maker.reset_free()
        while True:
            scale = penalty

            def penalised(hypo_maker, metric, scale=scale):
                extra = 0.0 if external_priors_penalty is None else external_priors_penalty(hypo_maker=hypo_maker, metric=metric)
                return sign * scale * violation(hypo_maker.params) + extra

            for name, value in method_kwargs.get("starting_values", {}).items():
                self._update_values(hypo_maker, [_with_value(hypo_maker.params[name], value)])
            res = self._inner(data_dist, hypo_maker, metric, penalised, local_fit_kwargs, *args)
            penalty *= 2
            if violation(res.params) <= tol:
                return res

    def _fit_ranges(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                    store_fit_history, include_metric_maps):
        """the inner fit with one parameter confined to each of several ranges in turn, started at the same
        relative position in each; the best kept (analysis.py:1396-1495)"""
        name = method_kwargs["param_name"]
        args = (data_dist, hypo_maker, metric, external_priors_penalty, local_fit_kwargs, store_fit_history,
                include_metric_maps)
        if name not in hypo_maker.params.free.names:
            return self._inner(*args)
        original = deepcopy(hypo_maker.params[name])
        rescaled = original._rescaled_value  # pylint: disable=protected-access
        results = []
        for interval in method_kwargs["ranges"]:
            moved = deepcopy(original)
            moved.range = interval                      # (a range is not checked against the value it finds)
            moved._rescaled_value = rescaled  # pylint: disable=protected-access
            moved.nominal_value = moved.value
            self._update_values(hypo_maker, moved, update_range=True, update_nominal_values=True)
            results.append(self._inner(*args))
        sign = self._sign(metric)
        best = results[int(np.argmin([sign * r.metric_val for r in results]))]
        best.params[name].range = original.range
        best.params[name].nominal_value = original.nominal_value
        self._update_values(hypo_maker, list(best.params.free), update_range=True, update_nominal_values=True)
        return best

    def _fit_staged(self, data_dist, hypo_maker, metric, external_priors_penalty, method_kwargs, local_fit_kwargs,
                    store_fit_history, include_metric_maps):
        """sub-fits one after the other, each starting where the one before ended: the nominal values are moved
        to the last best fit so that an inner `reset_free` keeps the progress (analysis.py:1497-1559)"""
        assert isinstance(local_fit_kwargs, list) and len(local_fit_kwargs) > 1
        nominal = {p.name: p.nominal_value for p in hypo_maker.params.free}
        best = None
        for spec in local_fit_kwargs:
            if best is not None:
                self._update_values(hypo_maker, list(best.params.free), update_nominal_values=True)
            best = self._inner(data_dist, hypo_maker, metric, external_priors_penalty, spec, store_fit_history,
                               include_metric_maps)
            for p in best.params.free:
                p.nominal_value = p.value
        for p in best.params.free:
            p.nominal_value = nominal[p.name]
        self._update_values(hypo_maker, list(best.params.free), update_nominal_values=True)
        return best


def _with_value(param, value):
    p = deepcopy(param)
    p.value = value
    return p
