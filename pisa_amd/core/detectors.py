"""`Detectors`: several `DistributionMaker`s, one per detector, fitted together (counterpart of
pisa/core/detectors.py:36-381).  Pipelines are grouped by their `detector_name`; a parameter two detectors
have under one name is ONE parameter of the fit only if it is listed in `shared_params`, otherwise the second
detector's appears as `<name>_<detector_name>`.  `get_outputs` returns one entry per detector."""
from collections import OrderedDict
from copy import deepcopy

import numpy as np

from pisa_amd.core.config_parser import PISAConfigParser
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.param import Param, ParamSet
from pisa_amd.core.pipeline import Pipeline

__all__ = ["Detectors"]


class Detectors:
    def __init__(self, pipelines, label=None, set_livetime_from_data=True, profile=False, shared_params=None):
        self.label = label
        self._profile = profile
        self.shared_params = list(shared_params) if shared_params is not None else []
        if isinstance(pipelines, (str, PISAConfigParser, OrderedDict, Pipeline)):
            pipelines = [pipelines]
        groups, self.det_names = [], []
        for pipeline in pipelines:
            if not isinstance(pipeline, Pipeline):
                pipeline = Pipeline(pipeline, profile=profile)
            name = pipeline.detector_name
            if name in self.det_names:
                groups[self.det_names.index(name)].append(pipeline)
            else:
                groups.append([pipeline])
                self.det_names.append(name)
        if None in self.det_names and len(self.det_names) > 1:
            raise NameError("At least one of the used pipelines has no detector_name.")
        self._distribution_makers = [DistributionMaker(g, set_livetime_from_data=set_livetime_from_data, profile=profile)
                                     for g in groups]
        for sp in self.shared_params:
            have = sum(sp in d.params.names for d in self)
            free = sum(sp in d.params.free.names for d in self)
            if have < 2:
                raise NameError("Shared param %s only exists in %d detectors." % (sp, have))
            if 0 < free != have:
                raise NameError("Shared param %s exists in %d detectors but only a free param in %d detectors."
                                % (sp, have, free))
        self._params_hash = None
        self.init_params()

    def __iter__(self):
        return iter(self._distribution_makers)

    def __repr__(self):
        return "Detectors(%s)" % ", ".join("%s: %d pipeline(s)" % (d.detector_name, len(d.pipelines)) for d in self)

    distribution_makers = property(lambda self: self._distribution_makers)
    params = property(lambda self: self._params)

    @property
    def profile(self):
        return self._profile

    @profile.setter
    def profile(self, value):
        for d in self:
            d.profile = value
        self._profile = value

    def report_profile(self, detailed=False, **kwargs):
        for d in self:
            print("%s:" % d.detector_name)
            d.report_profile(detailed=detailed)

    def run(self):
        for d in self:
            d.run()

    def setup(self):
        for d in self:
            d.setup()

    def get_outputs(self, **kwargs):
        """one entry per detector: a MapSet (`return_sum=True`) or the list of its pipelines' MapSets.  Values set
        on `self.params` since the last call are handed to the detectors first (detectors.py:149-168)."""
        h = self._params.hash
        if h != self._params_hash:
            self.update_params(self._params, init_params=False)
            self._params_hash = h
        return [d.get_outputs(**kwargs) for d in self]

    def update_params(self, params, init_params=True):
        """`params` into every detector; `<name>_<detector_name>` goes to that detector as `<name>`
        (detectors.py:170-194)"""
        if isinstance(params, Param):
            params = ParamSet(params)
        for d in self:
            ps = deepcopy(params)
            for name in list(ps.names):
                if d.detector_name is not None and name.endswith("_" + d.detector_name):
                    plain = name[: -len("_" + d.detector_name)]
                    if plain in ps.names:
                        ps.remove(plain)
                    renamed = ps[name]
                    renamed.name = plain
                    ps._reindex()
            d.update_params(ps)
        if init_params:
            self.init_params()

    def select_params(self, selections, error_on_missing=True):
        for d in self:
            d.select_params(selections, error_on_missing=error_on_missing)
        self.init_params()

    def init_params(self):
        """the fit's parameter set: the shared params first, then every detector's own; a name that is already
        taken (and not shared) gets the detector's name appended (detectors.py:209-237)"""
        params = ParamSet()
        for name in self.shared_params:
            for d in self:
                if name in d.params.names:
                    params.extend(d.params[name])
                    break
        for d in self:
            for prm in d.params:
                if prm.name in self.shared_params:
                    continue
                if prm.name in params.names:
                    twin = deepcopy(prm)
                    twin.name = "%s_%s" % (prm.name, d.detector_name)
                    params.extend(twin)
                else:
                    params.extend(prm)
        self._params = params
        self._params_hash = params.hash

    @property
    def shared_param_ind_list(self):
        """per detector: (position among its free params, position in `shared_params`) of its shared free params"""
        out = []
        for d in self:
            free = list(d.params.free.names)
            out.append([(free.index(n), self.shared_params.index(n)) for n in free if n in self.shared_params])
        return out if self.shared_params else []

    @property
    def param_selections(self):
        selections = None
        for d in self:
            mine = sorted(d.param_selections)
            if selections is not None and mine != selections:
                raise AssertionError("Different param_selections for different detectors.")
            selections = mine
        return selections

    @property
    def num_events_per_bin(self):
        return [d.num_events_per_bin for d in self]

    @property
    def empty_bin_indices(self):
        return [np.where(n == 0)[0] for n in self.num_events_per_bin]

    def set_free_params(self, values):
        """values in the order of `self.params.free` (detectors.py:306-324)"""
        mine = dict(zip(self._params.free.names, values))
        for d in self:
            vals = []
            for name in d.params.free.names:
                own = "%s_%s" % (name, d.detector_name)
                vals.append(mine[own] if own in mine else mine[name])
            d.set_free_params(vals)
        self.init_params()

    def randomize_free_params(self, random_state=None):
        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        self._set_rescaled_free_params(rs.rand(len(self._params.free)))

    def reset_all(self):
        for d in self:
            d.reset_all()
        self.init_params()

    def reset_free(self):
        for d in self:
            d.reset_free()
        self.init_params()

    def set_nominal_by_current_values(self):
        for d in self:
            d.set_nominal_by_current_values()
        self.init_params()

    def _set_rescaled_free_params(self, rvalues):
        """[0, 1] values in the order of `self.params.free`: the shared ones first, then each detector's own in
        its order (detectors.py:353-381)"""
        rvalues = list(rvalues)
        n_shared_free = len([n for n in self.shared_params if n in self._params.free.names])
        shared = [rvalues.pop(0) for _ in range(n_shared_free)] if self.shared_params else []
        spi = self.shared_param_ind_list
        shared_free_names = [n for n in self.shared_params if n in self._params.free.names]
        for i, d in enumerate(self):
            n_free = len(d.params.free)
            slots = spi[i] if self.shared_params else []
            own = [rvalues.pop(0) for _ in range(n_free - len(slots))]
            for pos, which in sorted(slots):
                own.insert(pos, shared[shared_free_names.index(self.shared_params[which])])
            d._set_rescaled_free_params(own)
        self.init_params()
