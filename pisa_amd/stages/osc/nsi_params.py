"""Non-standard-interaction coupling matrices.

Host-side counterpart of pisa/stages/osc/nsi_params.py: `StdNSIParams`
(nsi_params.py:62-181) and `VacuumLikeNSIParams` (:184-385).  Only
`eps_matrix` is consumed by the hot path (prob3.py:549-553: mat_pot =
diag(1 | 1.02, 0, 0) + eps_matrix).
"""
import numpy as np

from pisa_amd import CTYPE, FTYPE

__all__ = ["NSIParams", "StdNSIParams", "VacuumLikeNSIParams"]

_ALLCLOSE = dict(rtol=1e-12, atol=np.finfo(FTYPE).eps, equal_nan=True)


def _magnitude_phase(value):
    """(magnitude, phase) validation of nsi_params.py:29-45."""
    try:
        magnitude, phase = value
    except Exception:
        raise ValueError("Pass an iterable with two items (magnitude and phase)!")
    if not np.isscalar(magnitude) or not np.isscalar(phase):
        raise TypeError("Only scalar values for magnitude and phase accepted!")
    if magnitude < 0.0 and phase != 0.0:
        raise ValueError("Only accepting negative values with a zero phase (real coupling)!")
    return magnitude, phase


def _finish(eps):
    """Remove the mu-mu entry from the diagonal (trace is unobservable), force a
    real diagonal and check Hermiticity (nsi_params.py:168-181)."""
    eps = eps - eps[1, 1] * np.eye(3, dtype=FTYPE)
    for i in range(3):
        eps[i, i] = eps[i, i].real + 0 * 1.0j
    assert np.allclose(eps, eps.conj().T, **_ALLCLOSE)
    return eps


class NSIParams:
    def __init__(self):
        self._eps_matrix = np.zeros((3, 3), dtype=CTYPE)


def _diag_prop(i, name):
    def get(self):
        return self.eps_matrix[i, i].real

    def set_(self, value):
        if isinstance(value, complex) or not np.isscalar(value):
            raise TypeError("%s must be a real number!" % name)
        self._eps_matrix[i, i] = value + 1.0j * self._eps_matrix[i, i].imag

    return property(get, set_)


def _offdiag_prop(i, j):
    def get(self):
        return self.eps_matrix[i, j]

    def set_(self, value):
        magnitude, phase = _magnitude_phase(value)
        self._eps_matrix[i, j] = magnitude * (np.cos(phase) + 1.0j * np.sin(phase))
        self._eps_matrix[j, i] = np.conjugate(self._eps_matrix[i, j])

    return property(get, set_)


class StdNSIParams(NSIParams):
    eps_ee = _diag_prop(0, "eps_ee")
    eps_mumu = _diag_prop(1, "eps_mumu")
    eps_tautau = _diag_prop(2, "eps_tautau")
    eps_emu = _offdiag_prop(0, 1)
    eps_etau = _offdiag_prop(0, 2)
    eps_mutau = _offdiag_prop(1, 2)

    @property
    def eps_matrix(self):
        return _finish(self._eps_matrix)


class VacuumLikeNSIParams(NSIParams):
    """eps = Q U diag(eps_scale, eps_prime, 0) U^+ Q^+ - mumu - diag(1,0,0)
    (nsi_params.py:326-385)."""

    def __init__(self):
        super().__init__()
        self.eps_scale = 1.0
        self.eps_prime = 0.0
        self.phi12 = self.phi13 = self.phi23 = 0.0
        self.alpha1 = self.alpha2 = 0.0
        self.deltansi = 0.0

    @staticmethod
    def _phase(x):
        return complex(np.cos(x), np.sin(x))

    @property
    def eps_matrix(self):
        ph = self._phase
        Qrel = np.array([ph(self.alpha1), ph(self.alpha2), ph(-(self.alpha1 + self.alpha2))]) \
            * np.eye(3, dtype=FTYPE)
        c12, s12 = np.cos(self.phi12), np.sin(self.phi12)
        c13, s13 = np.cos(self.phi13), np.sin(self.phi13)
        c23, s23 = np.cos(self.phi23), np.sin(self.phi23)
        R12 = np.array([[c12, s12, 0], [-s12, c12, 0], [0, 0, 1]], dtype=FTYPE)
        R13 = np.array([[c13, 0, s13], [0, 1, 0], [-s13, 0, c13]], dtype=FTYPE)
        R23 = np.array([[1, 0, 0], [0, c23, s23 * ph(-self.deltansi)],
                        [0, -s23 * ph(self.deltansi), c23]])
        Umat = np.matmul(R12, np.matmul(R13, R23))
        Dmat = np.array([self.eps_scale, self.eps_prime, 0], dtype=FTYPE) * np.eye(3, dtype=FTYPE)
        pot = np.matmul(Qrel, np.matmul(Umat, np.matmul(Dmat, np.matmul(Umat.conj().T, Qrel.conj().T))))
        pot = pot - pot[1, 1] * np.eye(3, dtype=FTYPE)
        pot[0, 0] = pot[0, 0] - 1.0
        for i in range(3):
            pot[i, i] = pot[i, i].real + 0 * 1.0j
        assert np.allclose(pot, pot.conj().T, **_ALLCLOSE)
        return pot

    eps_ee = property(lambda self: self.eps_matrix[0, 0].real)
    eps_emu = property(lambda self: self.eps_matrix[0, 1])
    eps_etau = property(lambda self: self.eps_matrix[0, 2])
    eps_mumu = property(lambda self: self.eps_matrix[1, 1].real)
    eps_mutau = property(lambda self: self.eps_matrix[1, 2])
    eps_tautau = property(lambda self: self.eps_matrix[2, 2].real)
