"""`DistributionMaker`: a list of pipelines whose output maps are summed
(counterpart of pisa/core/distribution_maker.py:53-520; `get_outputs`
:251-294, free-parameter rescaling :462)."""
from collections.abc import Sequence

from pisa_amd.core.map import MapSet
from pisa_amd.core.param import ParamSet
from pisa_amd.core.pipeline import Pipeline

__all__ = ["DistributionMaker"]


class DistributionMaker:
    def __init__(self, pipelines, label=None, profile=False):
        if isinstance(pipelines, (str, Pipeline)) or not isinstance(pipelines, Sequence):
            pipelines = [pipelines]
        self.label = label
        self._pipelines = [p if isinstance(p, Pipeline) else Pipeline(p, profile=profile)
                           for p in pipelines]
        self._profile = profile

    pipelines = property(lambda self: self._pipelines)

    def __iter__(self):
        return iter(self._pipelines)

    @property
    def params(self):
        params = ParamSet()
        for p in self._pipelines:
            params.update(p.params, existing_must_match=False, extend=True)
        return params

    def select_params(self, selections, error_on_missing=True):
        for p in self._pipelines:
            p.select_params(selections, error_on_missing=False)

    def update_params(self, params):
        for p in self._pipelines:
            p.update_params(params)

    def set_free_params(self, values):
        free = self.params.free
        assert len(values) == len(free)
        for prm, v in zip(free, values):
            prm.value = v

    def _set_rescaled_free_params(self, rvalues):
        """free params from their [0,1]-rescaled values (distribution_maker.py:462)"""
        free = self.params.free
        assert len(rvalues) == len(free)
        for prm, r in zip(free, rvalues):
            prm._rescaled_value = float(r)

    def randomize_free_params(self, random_state=None):
        self.params.randomize_free(random_state)

    def reset_free(self):
        self.params.reset_free()

    def get_outputs(self, return_sum=False, sum_map_name="total", **kwargs):
        outputs = [p.get_outputs(**kwargs) for p in self._pipelines]
        if return_sum:
            total = sum(sum(ms) for ms in outputs)   # distribution_maker.py:274-281
            total.name = sum_map_name
            return MapSet([total], name=self.label)
        return outputs
