"""Neutrino-decay parameter holder (pisa/stages/osc/decay_params.py:20-56)."""
import numpy as np

from pisa_amd import CTYPE

__all__ = ["DecayParams"]


class DecayParams:
    def __init__(self):
        self.decay_alpha3 = 0.0  # eV^2

    @property
    def decay_matrix(self):
        """diag(0, 0, -i alpha3) (decay_params.py:50-56)."""
        m = np.zeros((3, 3), dtype=CTYPE)
        m[2, 2] = 0 - self.decay_alpha3 * 1j
        return m
