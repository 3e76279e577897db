"""Vacuum oscillation parameters -> PMNS and mass-splitting matrices.

Host-side counterpart of pisa/stages/osc/osc_params.py (OscParams): same
attribute names and conventions (angles are stored as sin(theta), cos comes
from sqrt(1 - s^2), osc_params.py:178-184), so results are bit identical; the
matrices are 72+144 bytes and are rebuilt per parameter point on the host,
then travel in the kernel-argument block of the prob3 kernels.
"""
import math

import numpy as np

from pisa_amd import CTYPE, FTYPE

__all__ = ["OscParams"]


class OscParams:
    def __init__(self):
        self._s = {"12": 0.0, "13": 0.0, "23": 0.0, "14": 0.0}
        self._deltacp = 0.0
        self._trig_delta = (None, 0.0, 1.0)
        self.dm21 = 0.0
        self.dm31 = 0.0
        self.dm41 = 0.0

    # sinXY / thetaXY pairs (osc_params.py:86-152)
    def _get_sin(self, ij):
        return self._s[ij]

    def _set_sin(self, ij, value):
        assert abs(value) <= 1
        self._s[ij] = value

    sin12 = property(lambda self: self._get_sin("12"), lambda self, v: self._set_sin("12", v))
    sin13 = property(lambda self: self._get_sin("13"), lambda self, v: self._set_sin("13", v))
    sin23 = property(lambda self: self._get_sin("23"), lambda self, v: self._set_sin("23", v))
    sin14 = property(lambda self: self._get_sin("14"), lambda self, v: self._set_sin("14", v))
    theta12 = property(lambda self: np.arcsin(self.sin12),
                       lambda self, v: self._set_sin("12", np.sin(v)))
    theta13 = property(lambda self: np.arcsin(self.sin13),
                       lambda self, v: self._set_sin("13", np.sin(v)))
    theta23 = property(lambda self: np.arcsin(self.sin23),
                       lambda self, v: self._set_sin("23", np.sin(v)))
    theta14 = property(lambda self: np.arcsin(self.sin14),
                       lambda self, v: self._set_sin("14", np.sin(v)))

    @property
    def deltacp(self):
        return self._deltacp

    @deltacp.setter
    def deltacp(self, value):
        assert 0.0 <= value <= 2 * np.pi  # osc_params.py:167
        self._deltacp = value

    def _trig(self):
        """(s12, s13, s23, c12, c13, c23, sin delta, cos delta) as Python floats.  Python-float
        arithmetic is the IEEE arithmetic of numpy's float64 scalars (`**` is libm's pow in both,
        `math.sqrt` and `np.sqrt` are correctly rounded), at a third of the call overhead: the
        matrices below are bit-identical to the numpy-scalar formulation
        (tests/test_host_params.py) and are rebuilt at every point of a fit."""
        s12, s13, s23 = float(self.sin12), float(self.sin13), float(self.sin23)
        d = self.deltacp
        if self._trig_delta[0] != d:
            self._trig_delta = (d, float(np.sin(d)), float(np.cos(d)))
        return (s12, s13, s23, math.sqrt(1.0 - s12 ** 2), math.sqrt(1.0 - s13 ** 2),
                math.sqrt(1.0 - s23 ** 2), self._trig_delta[1], self._trig_delta[2])

    def mix_floats(self, reparam=False):
        """the nine entries of `mix_matrix_complex` (or `mix_matrix_reparam_complex`) as 18 Python floats
        (re, im interleaved, row major): the same arithmetic on the same operands, without building the array --
        what a fit loop writes straight into the kernels' parameter block"""
        s12, s13, s23, c12, c13, c23, sd, cd = self._trig()
        if reparam:
            return (c12 * c13, 0.0, s12 * c13 * cd, s12 * c13 * sd, s13, 0.0,
                    -s12 * c23 * cd - c12 * s23 * s13, s12 * c23 * sd,
                    c12 * c23 - s12 * s23 * s13 * cd, -s12 * s23 * s13 * sd, s23 * c13, 0.0,
                    s12 * s23 * cd - c12 * c23 * s13, -s12 * s23 * sd,
                    -c12 * s23 - s12 * c23 * s13 * cd, -s12 * c23 * s13 * sd, c23 * c13, 0.0)
        return (c12 * c13, 0.0, s12 * c13, 0.0, s13 * cd, -s13 * sd,
                -s12 * c23 - c12 * s23 * s13 * cd, -c12 * s23 * s13 * sd,
                c12 * c23 - s12 * s23 * s13 * cd, -s12 * s23 * s13 * sd, s23 * c13, 0.0,
                s12 * s23 - c12 * c23 * s13 * cd, -c12 * c23 * s13 * sd,
                -c12 * s23 - s12 * c23 * s13 * cd, -s12 * c23 * s13 * sd, c23 * c13, 0.0)

    def dm_floats(self):
        """the nine entries of `dm_matrix` as Python floats (row major)"""
        m0, m1, m2 = 0.0, float(self.dm21), float(self.dm31)
        delta = 5.0e-9
        if m1 == 0.0:
            m0 -= delta
        if m2 == 0.0:
            m2 += delta
        d01, d02, d12 = m0 - m1, m0 - m2, m1 - m2
        return (0.0, d01, d02, -d01, 0.0, d12, -d02, -d12, 0.0)

    @property
    def mix_matrix_complex(self):
        """PDG parameterisation (osc_params.py:174-211)."""
        s12, s13, s23, c12, c13, c23, sd, cd = self._trig()
        return np.array([
            [complex(c12 * c13, 0.0), complex(s12 * c13, 0.0), complex(s13 * cd, -s13 * sd)],
            [complex(-s12 * c23 - c12 * s23 * s13 * cd, -c12 * s23 * s13 * sd),
             complex(c12 * c23 - s12 * s23 * s13 * cd, -s12 * s23 * s13 * sd), complex(s23 * c13, 0.0)],
            [complex(s12 * s23 - c12 * c23 * s13 * cd, -c12 * c23 * s13 * sd),
             complex(-c12 * s23 - s12 * c23 * s13 * cd, -s12 * c23 * s13 * sd), complex(c23 * c13, 0.0)],
        ], dtype=CTYPE)

    @property
    def mix_matrix(self):
        m = self.mix_matrix_complex
        return np.stack([m.real, m.imag], axis=2)

    @property
    def mix_matrix_reparam_complex(self):
        """diag(e^{i delta},1,1) U diag(e^{-i delta},1,1) (osc_params.py:213-258)."""
        s12, s13, s23, c12, c13, c23, sd, cd = self._trig()
        return np.array([
            [complex(c12 * c13, 0.0), complex(s12 * c13 * cd, s12 * c13 * sd), complex(s13, 0.0)],
            [complex(-s12 * c23 * cd - c12 * s23 * s13, s12 * c23 * sd),
             complex(c12 * c23 - s12 * s23 * s13 * cd, -s12 * s23 * s13 * sd), complex(s23 * c13, 0.0)],
            [complex(s12 * s23 * cd - c12 * c23 * s13, -s12 * s23 * sd),
             complex(-c12 * s23 - s12 * c23 * s13 * cd, -s12 * c23 * s13 * sd), complex(c23 * c13, 0.0)],
        ], dtype=CTYPE)

    @property
    def mix_matrix_reparam(self):
        m = self.mix_matrix_reparam_complex
        return np.stack([m.real, m.imag], axis=2)

    @property
    def dm_matrix(self):
        """dm[i, j] = m_i - m_j with the degeneracy nudges of osc_params.py:265-292."""
        m0, m1, m2 = 0.0, float(self.dm21), float(self.dm31)
        delta = 5.0e-9
        if m1 == 0.0:
            m0 -= delta
        if m2 == 0.0:
            m2 += delta
        d01, d02, d12 = m0 - m1, m0 - m2, m1 - m2
        return np.array([[0.0, d01, d02], [-d01, 0.0, d12], [-d02, -d12, 0.0]], dtype=FTYPE)
