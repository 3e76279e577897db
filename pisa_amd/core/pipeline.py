"""`Pipeline`: ordered list of services sharing one `ContainerSet`
(counterpart of pisa/core/pipeline.py:73-785).

`Pipeline(cfg)` parses the reference's cfg grammar, instantiates
`pisa_amd.stages.<stage>.<service>` (falling back to `<stage>.<service>` on
`sys.path`, like pipeline.py:282-296), shares same-named params between stages
(:342-346), applies `param_selections`, runs `setup()`; `get_outputs()` runs
every stage (`run` = `compute` + `apply`) and returns the output `MapSet`.
"""
from collections import OrderedDict
from importlib import import_module
from time import time

import numpy as np

from pisa_amd.core.binning import MultiDimBinning, OneDimBinning, VarBinning
from pisa_amd.core.config_parser import PISAConfigParser, parse_pipeline_config
from pisa_amd.core.container import Container, ContainerSet
from pisa_amd.core.param import ParamSet
from pisa_amd.core.stage import Stage

__all__ = ["Pipeline"]


class Pipeline:
    def __init__(self, config, profile=False):
        if isinstance(config, (str, PISAConfigParser)):
            config = parse_pipeline_config(config=config)
        elif not isinstance(config, OrderedDict):
            raise TypeError("`config` passed is of type %s but must be string, PISAConfigParser, "
                            "or OrderedDict" % type(config).__name__)
        self.name = config["pipeline"]["name"]
        self.detector_name = config["pipeline"].get("detector_name")
        self._data = ContainerSet(self.name)
        self._data["output_binning"] = config["pipeline"]["output_binning"]
        self.output_key = config["pipeline"]["output_key"]
        self._profile = profile
        self._plan = None               # core/fastplan.py: replay of the fused evaluation
        self._containers_stale = False  # the plan ran since the containers were last written
        self.fast_path = True
        self._setup_times, self._run_times, self._get_outputs_times = [], [], []
        self._stages = []
        self._config = config
        self._init_stages()

    # -- construction ---------------------------------------------------------------
    def _init_stages(self):
        stages = []
        for name, settings in self._config.items():
            if name == "pipeline":
                continue
            stage_name, service_name = name
            service_name = service_name.replace("pi_", "")
            try:
                module = import_module("pisa_amd.stages.%s.%s" % (stage_name, service_name))
            except ImportError:
                module = import_module("%s.%s" % (stage_name, service_name))
            service_cls = getattr(module, service_name)
            service = service_cls(**settings, profile=self._profile)
            if not isinstance(service, Stage):
                raise TypeError('Trying to create service "%s" for stage "%s", but the class is not '
                                "a Stage" % (service_name, stage_name))
            stages.append(service)
        self._stages = stages
        # identical names -> identical Param objects (pipeline.py:342-346)
        self.update_params(self.params, existing_must_match=True, extend=False)
        selections = set()
        for s in stages:
            selections.update(s.param_selections)
        for s in stages:
            s.select_params(sorted(selections), error_on_missing=False)
        self.setup()
        if isinstance(self.output_binning, VarBinning):     # pipeline.py:119-121
            self.assert_varbinning_compat()
            self.assert_exclusive_varbinning()

    # -- container protocol -----------------------------------------------------------
    def __len__(self):
        return len(self._stages)

    def __iter__(self):
        return iter(self._stages)

    def __getitem__(self, idx):
        if isinstance(idx, str):
            for s in self._stages:
                if idx in (s.service_name, s.stage_name, "%s.%s" % (s.stage_name, s.service_name)):
                    return s
            raise KeyError(idx)
        return self._stages[idx]

    def index(self, stage_id):
        """number of the stage with that `stage_name` (or that number), pipeline.py:199-220"""
        assert isinstance(stage_id, (int, str))
        for stage_num, stage in enumerate(self._stages):
            if stage_id in [stage_num, stage.stage_name]:
                return stage_num
        raise ValueError('No stage "%s" found in the pipeline.' % stage_id)

    def __getattr__(self, attr):
        # only reached when normal lookup failed: a stage by its stage name (pipeline.py:240-247)
        if not attr.startswith("_"):
            for stage in self.__dict__.get("_stages", ()):
                if stage.stage_name == attr:
                    return stage
        raise AttributeError('"%s" is neither a stage in this pipeline nor an attribute/property of the `Pipeline` object.'
                             % attr)

    def tabulate(self, tablefmt="plain"):
        """one row per stage: number, class, modes, which functions it has, fixed / free parameters (pipeline.py:138-146)"""
        from tabulate import tabulate

        headers = ["stage number", "name", "calc_mode", "apply_mode", "has setup", "has compute", "has apply",
                   "# fixed params", "# free params"]
        table = [[i, s.__class__.__name__, s.calc_mode, s.apply_mode, s.has_setup, s.has_compute, s.has_apply,
                  len(s.params.fixed), len(s.params.free)] for i, s in enumerate(self._stages)]
        return tabulate(table, headers, tablefmt=tablefmt, colalign=["right"] + ["center"] * (len(headers) - 1))

    def __repr__(self):
        return self.tabulate(tablefmt="presto")

    def _repr_html_(self):
        return self.tabulate(tablefmt="html")

    @property
    def hash(self):
        """of the stages' classes, modes and parameter states (pipeline.py:676-680 hashes source code and stage states)"""
        from pisa_amd.utils.hash import hash_obj

        return hash_obj([(type(s).__module__, type(s).__name__, str(s.calc_mode), str(s.apply_mode), s.params.values_hash,
                          tuple(s.params.names), tuple(p.is_fixed for p in s.params)) for s in self._stages])

    stages = property(lambda self: list(self._stages))
    stage_names = property(lambda self: [s.stage_name for s in self._stages])
    service_names = property(lambda self: [s.service_name for s in self._stages])
    config = property(lambda self: self._config)

    @property
    def data(self):
        """the ContainerSet; if evaluations were replayed by the fast plan since the stages last
        wrote it, the stages are run first so that a reader sees the current parameters' data"""
        if self._containers_stale:
            self.run()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    @property
    def output_binning(self):
        return self._data["output_binning"]

    @output_binning.setter
    def output_binning(self, binning):
        if isinstance(binning, VarBinning):     # checked against the events at hand; no new set-up (pipeline.py:767-771)
            self.assert_varbinning_compat()
            self.assert_exclusive_varbinning(output_binning=binning)
            self._data["output_binning"] = binning
            self._plan = None
            return
        self._data["output_binning"] = binning
        self.setup()

    @property
    def profile(self):
        return self._profile

    @profile.setter
    def profile(self, value):
        self._profile = bool(value)
        self._plan = None
        for s in self._stages:
            s.profile = self._profile

    # -- params ----------------------------------------------------------------------
    @property
    def params(self):
        """merged ParamSet of all stages (same Param objects); rebuilt only when a stage's
        selected parameter objects changed (select_params / update_params)"""
        hit = self.__dict__.get("_params_cache")
        if hit is not None and hit[0] == ParamSet.struct_clock:
            return hit[1]
        params = ParamSet()
        object.__setattr__(params, "_transient", True)   # a view: filling it changes no owned set
        for s in self._stages:
            params.update(s.params, existing_must_match=False, extend=True)
        self.__dict__["_params_cache"] = (ParamSet.struct_clock, params)
        return params

    @property
    def param_selections(self):
        sel = set()
        for s in self._stages:
            sel.update(s.param_selections)
        return sorted(sel)

    def update_params(self, params, existing_must_match=False, extend=False):
        for s in self._stages:
            s._param_selector.update(params, existing_must_match=existing_must_match, extend=extend)

    def add_covariance(self, covmat):
        """Correlated priors between parameters of this pipeline (pipeline.py:485-536; `ParamSet.add_covariance`):
        the correlated parameters become `DerivedParam`s in every stage that has them, and the new, uncorrelated
        `<name>_rotated` parameters -- the ones a fit moves -- join the first stage that holds a correlated one."""
        from pisa_amd.core.param import DerivedParam

        if self.__dict__.get("_covariance_set"):
            raise ValueError("A covariance matrix has been added already; add ONE larger matrix rather than calling"
                             " this several times")
        paramset = ParamSet(list(self.params))
        paramset.add_covariance(covmat)
        self._covariance_set = True
        self.update_params(paramset)                    # the DerivedParams replace their namesakes
        return self._add_rotated(paramset)

    def _add_rotated(self, paramset, suppress_warning=False):
        """the uncorrelated parameters of `paramset`'s DerivedParams into the first stage that has one of the
        derived ones (pipeline.py:507-536)"""
        from pisa_amd.core.param import DerivedParam

        derived = [p for p in paramset if isinstance(p, DerivedParam)]
        if not derived:
            return False
        rotated = list(derived[0].dependson.values())
        for s in self._stages:
            if any(d.name in s._param_selector.params.names for d in derived):
                s._param_selector.update(rotated, extend=True)
                break
        self._plan = None
        return True

    def select_params(self, selections, error_on_missing=False):
        found = False
        for s in self._stages:
            try:
                s.select_params(selections, error_on_missing=True)
                found = True
            except KeyError:
                pass
        if not found and error_on_missing:
            raise KeyError("None of the stages has all selections %s" % (selections,))

    # -- execution --------------------------------------------------------------------
    def setup(self):
        t0 = time()
        self._plan, self._containers_stale = None, False
        output_binning = self._data["output_binning"]
        self._data = ContainerSet(self.name)
        self._data["output_binning"] = output_binning
        for s in self._stages:
            s.data = self._data
            s.setup()
        if self._profile:
            self._setup_times.append(time() - t0)

    def run(self):
        t0 = time()
        self._containers_stale = False
        modes = [s.apply_mode for s in self._stages]
        if modes != self.__dict__.get("_apply_modes") and isinstance(self.output_binning, VarBinning):
            self.assert_varbinning_compat()     # a stage's apply_mode was changed between runs (pipeline.py:539-543)
        self._apply_modes = modes
        for s in self._stages:
            s.run()
        if self._profile:
            self._run_times.append(time() - t0)

    def get_outputs(self, output_binning=None, output_key=None):
        t0 = time()
        default = output_binning is None and output_key is None
        if default and self.fast_path and not self._profile:
            if self._plan is not None:
                try:
                    outputs = self._plan.run()
                except BaseException:
                    # the replay marks parameter changes as seen before it has applied all of them:
                    # a plan that raised half way (a kernel status, a value out of range in a stage's
                    # compute, an allocation) must not be replayed with its stale tables -- the next
                    # evaluation takes the ordinary Stage path with every bypassed memo invalidated
                    plan, self._plan = self._plan, None
                    plan.invalidate()
                    raise
                if outputs is not None:
                    return outputs
                self._plan = None
        self.run()
        if output_binning is None:
            output_binning = self.output_binning
        elif isinstance(output_binning, VarBinning):
            self.assert_exclusive_varbinning(output_binning=output_binning)
        if output_key is None:
            output_key = self.output_key
        if isinstance(output_binning, VarBinning):
            self.assert_varbinning_compat()
            outputs = self._get_outputs_varbinning(output_binning, output_key)
            if self._profile:
                self._get_outputs_times.append(time() - t0)
            return outputs
        assert isinstance(output_binning, MultiDimBinning)
        self._data.representation = output_binning
        if isinstance(output_key, tuple):
            assert len(output_key) == 2
            outputs = self._data.get_mapset(output_key[0], error=output_key[1])
        else:
            outputs = self._data.get_mapset(output_key)
        if default and self.fast_path and not self._profile:
            from pisa_amd.core.fastplan import FastPlan

            self._plan = FastPlan.build(self)
        if self._profile:
            self._get_outputs_times.append(time() - t0)
        return outputs

    # -- one MapSet per event selection (VarBinning) ----------------------------------
    def assert_varbinning_compat(self):
        """every stage that applies anything applies it to events: no histogramming service (pipeline.py:686-712)"""
        bad = [s for s in self._stages if s.apply_mode is not None and s.apply_mode != "events"]
        if bad:
            raise ValueError("When a variable binning is used, all stages need to set apply_mode='events', but '%s'"
                             " of '%s' do(es) not!" % (", ".join("%s.%s" % (s.stage_name, s.service_name) for s in bad),
                                                       self.name))

    def assert_exclusive_varbinning(self, output_binning=None):
        """no event of any container passes two of the cut expressions (pipeline.py:714-763); the bins of a
        `OneDimBinning` of selections are exclusive by construction"""
        vb = self.output_binning if output_binning is None else output_binning
        if not isinstance(vb, VarBinning) or not isinstance(vb.selections, list):
            return
        self._data.representation = "events"
        for c in self._data:
            passed = np.zeros(c.size, dtype=np.int64)
            for sel in vb.selections:
                passed += np.broadcast_to(np.asarray(c.get_keep_mask(sel), dtype=bool), passed.shape)
            if np.any(passed > 1):
                raise ValueError("Selections %s are not mutually exclusive for '%s' (at least) in pipeline '%s'!"
                                 % (vb.selections, c.name, self.name))

    def _selection_rows(self, container, selections, i):
        """device indices of the events of `container` in selection `i`; kept while the container's event
        columns are the same objects (the columns a cut reads do not move during a fit)"""
        import torch

        from pisa_amd import kernels as K

        if isinstance(selections, OneDimBinning):
            names = (selections.name,)
            key = (selections.hash, i)
        else:
            names = tuple(k for k in container.keys if k in selections[i])
            key = (selections[i], i)
        stamp = tuple(container.version(n) for n in names) + (container.size,)
        cache = container.__dict__.setdefault("_selection_rows", {})
        hit = cache.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        if isinstance(selections, OneDimBinning):
            var = container[selections.name]
            e = selections.edge_magnitudes
            keep = (var >= e[i]) & (var < e[i + 1])
        else:
            keep = np.broadcast_to(np.asarray(container.get_keep_mask(selections[i]), dtype=bool), (container.size,))
        rows = torch.from_numpy(np.flatnonzero(keep)).to(K.device())
        cache[key] = (stamp, rows)
        return rows

    def _get_outputs_varbinning(self, output_binning, output_key):
        """list of MapSets, one per selection: the selected events of every container histogrammed in that
        selection's binning, sum of weights and (with an error key) square root of the sum of their squares
        (pipeline.py:389-451).  The selected rows are gathered on the device; the histograms are the
        `array_to_binned` translation of a fresh container, as in the reference."""
        import torch

        self._data.representation = "events"
        key, err = output_key if isinstance(output_key, tuple) else (output_key, None)
        outputs = []
        for i, binning in enumerate(output_binning.binnings):
            containers = []
            for c in self._data:
                rows = self._selection_rows(c, output_binning.selections, i)
                cc = Container(c.name)
                for var in binning.names:
                    cc[var] = c.device(var).index_select(0, rows)
                w = c.device(key).index_select(0, rows)
                cc[key] = w
                cc.translation_modes[key] = "sum"
                if err is not None:
                    cc[err] = torch.square(w)
                    cc.translation_modes[err] = "sum"
                containers.append(cc)
            dat = ContainerSet(self._data.name, containers=containers, representation=binning)
            if err is not None:
                for cc in dat:
                    cc[err] = torch.sqrt(cc.device(err))
                outputs.append(dat.get_mapset(key, error=err))
            else:
                outputs.append(dat.get_mapset(key))
        return outputs

    def report_profile(self, detailed=False):
        for label, times in (("setup", self._setup_times), ("run", self._run_times),
                             ("get_outputs", self._get_outputs_times)):
            if times:
                print("%-12s total %.5f s, n=%d, mean %.5f s" % (label, np.sum(times), len(times),
                                                               np.mean(times)))
        for s in self._stages:
            s.report_profile(detailed=detailed)
