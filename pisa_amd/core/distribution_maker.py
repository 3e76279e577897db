"""`DistributionMaker`: a list of pipelines whose output maps are summed
(counterpart of pisa/core/distribution_maker.py:53-520; `get_outputs`
:251-294, free-parameter rescaling :462)."""
from collections.abc import Mapping, Sequence

import numpy as np

from pisa_amd.core.map import MapSet
from pisa_amd.core.param import Param, ParamSet
from pisa_amd.core.pipeline import Pipeline

__all__ = ["DistributionMaker"]


class DistributionMaker:
    def __init__(self, pipelines, label=None, set_livetime_from_data=True, profile=False):
        if isinstance(pipelines, (str, Pipeline)) or not isinstance(pipelines, Sequence):
            pipelines = [pipelines]
        self.label = label
        self._pipelines = [p if isinstance(p, Pipeline) else Pipeline(p, profile=profile)
                           for p in pipelines]
        self._profile = profile
        self.metadata = {}
        if set_livetime_from_data:
            self._livetime_from_data()
        # all pipelines of a maker belong to one detector (distribution_maker.py:178-187)
        self.detector_name = "no_name"
        for p in self._pipelines:
            if p.detector_name != self.detector_name and self.detector_name != "no_name":
                raise NameError("Different detector names in distribution_maker pipelines")
            self.detector_name = p.detector_name
        self._unify_params()

    def _livetime_from_data(self):
        """a stage that carries `metadata['livetime']` (a data loader) fixes `params.livetime` of every pipeline that
        has it, in seconds (distribution_maker.py:113-171)"""
        from pisa_amd.core.units import ureg

        found = None
        for ip, p in enumerate(self._pipelines):
            for js, s in enumerate(p.stages):
                md = getattr(s, "metadata", None)
                if not (isinstance(md, Mapping) and "livetime" in md):
                    continue
                if found is None:
                    found = md["livetime"]
                if md["livetime"] != found:
                    raise ValueError("Pipeline index %d, stage index %d has data livetime = %s, in disagreement with"
                                     " previously-found livetime = %s" % (ip, js, md["livetime"], found))
        self.metadata["livetime"] = found
        if found is None:
            return
        livetime = found * ureg.sec
        for p in self._pipelines:
            if "livetime" in p.params.names:
                p.params.livetime.is_fixed = True
                if p.params.livetime.value != livetime:
                    p.params.livetime = livetime

    @property
    def profile(self):
        return self._profile

    @profile.setter
    def profile(self, value):
        for p in self._pipelines:
            p.profile = value
        self._profile = value

    def report_profile(self, detailed=False, **kwargs):
        for p in self._pipelines:
            p.report_profile(detailed=detailed)

    @property
    def num_events_per_bin(self):
        """unweighted events of all pipelines per output bin, flat (distribution_maker.py:385-408)"""
        from pisa_amd.core.translation import histogram

        binning = self._pipelines[0].output_binning
        total = np.zeros(binning.size)
        for p in self._pipelines:
            assert p.output_binning == binning
            data = p.data
            keep = data.representation
            try:
                data.representation = "events"
                for c in data:
                    if all(n in c.keys for n in binning.names) and c.size:
                        total += histogram([c[n] for n in binning.names], None, binning, averaged=False)
            finally:
                data.representation = keep
        return total

    @property
    def empty_bin_indices(self):
        return np.where(self.num_events_per_bin == 0)[0]

    def tabulate(self, tablefmt="plain"):
        """one row per pipeline (distribution_maker.py:203-218)"""
        from tabulate import tabulate

        headers = ["pipeline number", "name", "detector name", "output_binning", "output_key", "profile"]
        table = [[i, p.name, p.detector_name, getattr(p.output_binning, "name", None), p.output_key, p.profile]
                 for i, p in enumerate(self._pipelines)]
        return tabulate(table, headers, tablefmt=tablefmt, colalign=["right"] + ["center"] * (len(headers) - 1))

    def __repr__(self):
        return self.tabulate(tablefmt="presto")

    def _repr_html_(self):
        return self.tabulate(tablefmt="html")

    @property
    def hash(self):
        from pisa_amd.utils.hash import hash_obj

        return hash_obj([p.hash for p in self._pipelines])

    pipelines = property(lambda self: self._pipelines)

    def __iter__(self):
        return iter(self._pipelines)

    def _unify_params(self):
        """Same-named params become ONE object in every pipeline, for every selection
        (distribution_maker.py:183-196), so that a value set through one pipeline -- or through
        the merged `self.params` -- is seen by all of them."""
        original = self.param_selections
        selections = set()
        for p in self._pipelines:
            for s in p.stages:
                selections.update(s._param_selector._selector_sets.keys())
        for sel in sorted(selections):
            self.select_params(sel)
            merged = self.params
            for p in self._pipelines:
                p.update_params(merged, existing_must_match=True, extend=False)
        if original:
            self.select_params(original)
        merged = self.params
        for p in self._pipelines:
            p.update_params(merged, existing_must_match=True, extend=False)

    def add_covariance(self, covmat):
        """correlated priors between parameters of any of the pipelines (distribution_maker.py:327-348)"""
        paramset = ParamSet(list(self.params))
        paramset.add_covariance(covmat)
        self.update_params(paramset)
        done = [p._add_rotated(paramset, suppress_warning=True) for p in self._pipelines]
        if not any(done):
            raise ValueError("no pipeline holds one of the correlated parameters")
        self.__dict__.pop("_params_view", None)

    @property
    def params(self):
        """merged view of the pipelines' parameters (distribution_maker.py:310-317); the Param objects
        are the pipelines' own, so the view is rebuilt only when some set changed structurally"""
        hit = getattr(self, "_params_view", None)
        if hit is not None and hit[0] == ParamSet.struct_clock:
            return hit[1]
        params = ParamSet()
        object.__setattr__(params, "_transient", True)
        for p in self._pipelines:
            params.update(p.params, existing_must_match=False, extend=True)
        self._params_view = (ParamSet.struct_clock, params)
        return params

    @property
    def param_selections(self):
        sel = set()
        for p in self._pipelines:
            sel.update(p.param_selections)
        return sorted(sel)

    def select_params(self, selections, error_on_missing=True):
        for p in self._pipelines:
            p.select_params(selections, error_on_missing=False)

    def update_params(self, params):
        for p in self._pipelines:
            p.update_params(params)

    def _for_each_free(self, values, setter):
        """distribution_maker.py:420-436, 462-476: every pipeline that has the parameter free gets
        the value; a parameter that is free somewhere and fixed elsewhere is an error."""
        names = self.params.free.names
        assert len(values) == len(names)
        for pipeline in self._pipelines:
            pp = pipeline.params
            free = set(pp.free.names)
            for name, value in zip(names, values):
                if name in free:
                    setter(pp[name], value)
                elif name in pp.names:
                    raise AttributeError('Trying to set value for "%s", a parameter that is fixed in '
                                         "at least one pipeline" % name)

    def set_free_params(self, values):
        self._for_each_free(values, lambda prm, v: setattr(prm, "value", v))

    def _set_rescaled_free_params(self, rvalues):
        """free params from their [0,1]-rescaled values (distribution_maker.py:462-476).  The list of
        parameter objects behind every free name (one per pipeline that has it free; after
        `_unify_params` usually ONE shared object) is kept until a set changes structurally or a
        parameter is fixed / freed: a minimiser calls this at every point."""
        key = (ParamSet.struct_clock, Param.fix_clock)
        c = getattr(self, "_free_targets", None)
        if c is None or c[0] != key:
            targets = [[] for _ in self.params.free.names]
            self._for_each_free(targets, lambda prm, t: t.append(prm) if all(prm is not q for q in t) else None)
            c = self._free_targets = (key, targets)
        assert len(rvalues) == len(c[1])
        for targets, r in zip(c[1], rvalues):
            r = float(r)
            for prm in targets:
                prm._rescaled_value = r

    def metric_many(self, rescaled_points, data_dist, metric, on_point=None):
        """[data_dist.metric_total(template at x, metric) + priors penalty at x  for x in points]:
        the template evaluated at several INDEPENDENT points of the [0,1]-rescaled free parameters.
        With one pipeline of the replayable shape whose moving parameters belong to osc.prob3 / aeff.aeff
        the points share one sweep of the events (`FastPlan.metric_many`); otherwise, and with identical
        results, point by point.  `on_point(i)` is called while the parameters sit at point i (a fit history
        reads the values there; it may be called again for a point if the sweep had to be abandoned).  The parameters are left at the last point."""
        import numpy as np

        from pisa_amd.core.map import Map

        pts = [np.clip(np.asarray(x, dtype=np.float64), 0.0, 1.0) for x in rescaled_points]
        if isinstance(data_dist, list):        # one MapSet per selection of a variable binning: point by point
            data_map = None
        else:
            data_map = data_dist if isinstance(data_dist, Map) else (data_dist.maps[0] if len(data_dist) == 1 else None)
        plan = self._pipelines[0]._plan if len(self._pipelines) == 1 and self._pipelines[0].fast_path else None
        if plan is not None and data_map is not None and len(pts) > 1:
            pens = []

            def set_point(i):
                self._set_rescaled_free_params(pts[i])
                pens.append(self.params.priors_penalty(metric=metric))
                if on_point is not None:
                    on_point(i)

            try:
                vals = plan.metric_many(set_point, len(pts), data_map.hist, metric)
            except BaseException:
                self._pipelines[0]._plan = None
                plan.invalidate()
                raise
            if vals is not None:
                return [v + p for v, p in zip(vals, pens)]
        out = []
        for i, x in enumerate(pts):
            self._set_rescaled_free_params(x)
            if on_point is not None:
                on_point(i)
            hypo = self.get_outputs(return_sum=True)
            if isinstance(hypo, list):
                val = sum(d.metric_total(expected_values=h, metric=metric) for d, h in zip(data_dist, hypo))
            else:
                val = data_dist.metric_total(expected_values=hypo, metric=metric)
            out.append(val + self.params.priors_penalty(metric=metric))
        return out

    def randomize_free_params(self, random_state=None):
        import numpy as np

        rs = random_state if isinstance(random_state, np.random.RandomState) \
            else np.random.RandomState(random_state)
        self._set_rescaled_free_params(rs.rand(len(self.params.free)))

    def reset_all(self):
        for p in self._pipelines:
            p.params.reset_all()

    def reset_free(self):
        for p in self._pipelines:
            p.params.reset_free()

    def set_nominal_by_current_values(self):
        for p in self._pipelines:
            p.params.set_nominal_by_current_values()

    def run(self):
        for p in self._pipelines:
            p.run()

    def setup(self):
        for p in self._pipelines:
            p.setup()

    def get_outputs(self, return_sum=False, sum_map_name="total", **kwargs):
        outputs = [p.get_outputs(**kwargs) for p in self._pipelines]
        if return_sum and isinstance(outputs[0], list):
            # pipelines with a VarBinning: one summed MapSet per selection (distribution_maker.py:283-291)
            outs = []
            for i in range(len(outputs[0])):
                total = sum(ms[i].total(sum_map_name) for ms in outputs)
                total.name = sum_map_name
                outs.append(MapSet([total], name=self.label))
            return outs
        if return_sum:
            # distribution_maker.py:274-281: sum([sum(x) for x in outputs]); `total()` is that sum
            # over one pipeline's maps -- for device-backed outputs without bringing them home
            total = sum(ms.total(sum_map_name) for ms in outputs)
            total.name = sum_map_name
            return MapSet([total], name=self.label)
        return outputs
