"""Long-range-interaction potential matrices (pisa/stages/osc/lri_params.py:31-108)."""
import numpy as np

from pisa_amd import FTYPE

__all__ = ["LRIParams"]


class LRIParams:
    def __init__(self):
        self._v_lri = 0.0

    @property
    def v_lri(self):
        return self._v_lri

    @v_lri.setter
    def v_lri(self, value):
        assert value < 1.0  # lri_params.py:48
        self._v_lri = value

    def _diag(self, plus, minus):
        v = np.zeros((3, 3), dtype=FTYPE)
        v[plus, plus] = self.v_lri
        v[minus, minus] = -self.v_lri
        return v

    potential_matrix_emu = property(lambda self: self._diag(0, 1))
    potential_matrix_etau = property(lambda self: self._diag(0, 2))
    potential_matrix_mutau = property(lambda self: self._diag(1, 2))
