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

    def __repr__(self):
        vals = ", ".join("%s=%s" % (p.name, p.value) for p in self.params.free)
        return "HypoFitResult(%s=%.8g; %s)" % (self.metric, self.metric_val, vals)


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
        method = ms["method"].lower()
        options = dict(ms.get("options", {}))
        if (batched_gradient and method in ("l-bfgs-b", "slsqp") and hasattr(hypo_maker, "metric_many")
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
                bounds=bounds if method in ("l-bfgs-b", "slsqp", "tnc", "trust-constr") else None,
                method=ms["method"], options=options)
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
