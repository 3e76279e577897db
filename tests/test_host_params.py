"""CPU tests of the PRODUCT's host-side parameter matrices (SURVEY §8 a1-a3:
`pisa_amd/stages/osc/{osc,nsi,decay,lri}_params.py`) against values produced by the
reference's own classes (`tests/golden/params_ref.npz`, made by `oracle/gen_golden.py:gen_params`
from pisa/stages/osc/osc_params.py:174-292, nsi_params.py:168-181, 326-385,
decay_params.py, lri_params.py).  Bit-exact: the same numpy operations in the same order."""
import numpy as np

from tests.conftest import load_golden


def test_osc_params_bit_exact():
    from pisa_amd.stages.osc.osc_params import OscParams

    g = load_golden("params_ref.npz")
    inputs = g["osc::inputs"]
    assert inputs.shape == (8, 6)
    for i, (t12, t13, t23, dcp, dm21, dm31) in enumerate(inputs):
        o = OscParams()
        o.theta12, o.theta13, o.theta23, o.deltacp, o.dm21, o.dm31 = t12, t13, t23, dcp, dm21, dm31
        np.testing.assert_array_equal(o.mix_matrix_complex, g["osc%d::mix" % i])
        np.testing.assert_array_equal(o.mix_matrix_reparam_complex, g["osc%d::mix_reparam" % i])
        np.testing.assert_array_equal(o.dm_matrix, g["osc%d::dm" % i])
        # unitarity as a sanity property of the golden itself
        u = o.mix_matrix_complex
        np.testing.assert_allclose(u @ u.conj().T, np.eye(3), atol=1e-15)


def test_nsi_params_bit_exact():
    from pisa_amd.stages.osc.nsi_params import StdNSIParams, VacuumLikeNSIParams

    g = load_golden("params_ref.npz")
    for i, v in enumerate(g["stdnsi::inputs"]):
        n = StdNSIParams()
        n.eps_ee, n.eps_emu, n.eps_etau = v[0], (v[1], v[2]), (v[3], v[4])
        n.eps_mumu, n.eps_mutau, n.eps_tautau = v[5], (v[6], v[7]), v[8]
        eps = n.eps_matrix
        np.testing.assert_array_equal(eps, g["stdnsi%d::eps" % i])
        np.testing.assert_array_equal(eps, eps.conj().T)  # Hermitian
    for i, v in enumerate(g["vacnsi::inputs"]):
        n = VacuumLikeNSIParams()
        (n.eps_scale, n.eps_prime, n.phi12, n.phi13, n.phi23, n.alpha1, n.alpha2, n.deltansi) = v
        np.testing.assert_array_equal(n.eps_matrix, g["vacnsi%d::eps" % i])


def test_decay_and_lri_params_bit_exact():
    from pisa_amd.stages.osc.decay_params import DecayParams
    from pisa_amd.stages.osc.lri_params import LRIParams

    g = load_golden("params_ref.npz")
    d = DecayParams()
    d.decay_alpha3 = float(g["decay::alpha3"])
    np.testing.assert_array_equal(d.decay_matrix, g["decay::matrix"])
    l = LRIParams()
    l.v_lri = float(g["lri::v"])
    np.testing.assert_array_equal(l.potential_matrix_emu, g["lri::emu"])
    np.testing.assert_array_equal(l.potential_matrix_etau, g["lri::etau"])
    np.testing.assert_array_equal(l.potential_matrix_mutau, g["lri::mutau"])


def test_tomography_scaling_arrays():
    """pisa/stages/osc/scaling_params.py:26-146 (the reference's module needs pint and cannot be
    imported here, so no golden): the constrained scaling must conserve the Earth's mass and moment
    of inertia for any core factor, reduce to ones at alpha = 1, and refuse negative factors."""
    from pisa_amd.stages.osc.scaling_params import (FIVE_LAYER_RADII, FIVE_LAYER_RHOS, Core_scaling_w_constrain,
                                                    Core_scaling_wo_constrain, Mass_scaling)

    r, rho = FIVE_LAYER_RADII, FIVE_LAYER_RHOS
    mass = np.array([rho[k] * (r[k] ** 3 - r[k - 1] ** 3) for k in range(1, 6)])       # x 4 pi / 3
    inertia = np.array([rho[k] * (r[k] ** 5 - r[k - 1] ** 5) for k in range(1, 6)])    # x 8 pi / 15
    c = Core_scaling_w_constrain()
    for alpha in (1.0, 0.9, 1.1, 1.25):
        c.core_density_scale = alpha
        s = c.scaling_array
        assert s.shape == (6,) and s[0] == 1.0 and np.all(s[3:] == alpha)
        centre_out = s[::-1][1:]      # [alpha (inner core), alpha (outer core), beta, gamma, 1]; s[5] is the r = 0 row
        np.testing.assert_allclose(np.dot(centre_out, mass), mass.sum(), rtol=1e-12)
        np.testing.assert_allclose(np.dot(centre_out, inertia), inertia.sum(), rtol=1e-12)
    c.core_density_scale = 1.0
    np.testing.assert_allclose(c.scaling_array, np.ones(6), rtol=1e-12)
    c.core_density_scale = 3.0
    import pytest

    with pytest.raises(AssertionError):
        c.scaling_array      # the mantle factors would be negative
    w = Core_scaling_wo_constrain()
    w.core_density_scale, w.innermantle_density_scale, w.middlemantle_density_scale = 1.2, 0.9, 1.05
    np.testing.assert_array_equal(w.scaling_factor_array, [1.0, 1.05, 0.9, 1.2, 1.2, 1.2])
    m = Mass_scaling()
    m.density_scale = 1.3
    assert m.density_scale == 1.3
    with pytest.raises(AssertionError):
        m.density_scale = -0.1


def test_mixing_matrices_equal_the_numpy_scalar_formulation_bit_for_bit():
    """`OscParams` does its scalar arithmetic with Python floats (a third of the call overhead of
    numpy scalars, and the matrices are rebuilt at every point of a fit); the reference does it with
    numpy float64 scalars (osc_params.py:174-258).  Same IEEE operations in the same order: the
    results must not differ in any bit, whatever the parameter values."""
    from pisa_amd.stages.osc.osc_params import OscParams

    def numpy_form(s12, s13, s23, dcp, reparam):
        c12, c13, c23 = np.sqrt(1.0 - s12 ** 2), np.sqrt(1.0 - s13 ** 2), np.sqrt(1.0 - s23 ** 2)
        sd, cd = np.sin(dcp), np.cos(dcp)
        if not reparam:
            re = np.array([[c12 * c13, s12 * c13, s13 * cd],
                           [-s12 * c23 - c12 * s23 * s13 * cd, c12 * c23 - s12 * s23 * s13 * cd, s23 * c13],
                           [s12 * s23 - c12 * c23 * s13 * cd, -c12 * s23 - s12 * c23 * s13 * cd, c23 * c13]])
            im = np.array([[0.0, 0.0, -s13 * sd],
                           [-c12 * s23 * s13 * sd, -s12 * s23 * s13 * sd, 0.0],
                           [-c12 * c23 * s13 * sd, -s12 * c23 * s13 * sd, 0.0]])
        else:
            re = np.array([[c12 * c13, s12 * c13 * cd, s13],
                           [-s12 * c23 * cd - c12 * s23 * s13, c12 * c23 - s12 * s23 * s13 * cd, s23 * c13],
                           [s12 * s23 * cd - c12 * c23 * s13, -c12 * s23 - s12 * c23 * s13 * cd, c23 * c13]])
            im = np.array([[0.0, s12 * c13 * sd, 0.0],
                           [s12 * c23 * sd, -s12 * s23 * s13 * sd, 0.0],
                           [-s12 * s23 * sd, -s12 * c23 * s13 * sd, 0.0]])
        return re + im * 1.0j

    rs = np.random.RandomState(12)
    o = OscParams()
    for _ in range(3000):
        th = rs.rand(3) * np.pi / 2
        dcp = rs.rand() * 2 * np.pi
        o.theta12, o.theta13, o.theta23, o.deltacp = th[0], th[1], th[2], dcp
        for reparam in (False, True):
            want = numpy_form(np.sin(th[0]), np.sin(th[1]), np.sin(th[2]), dcp, reparam)
            got = o.mix_matrix_reparam_complex if reparam else o.mix_matrix_complex
            assert got.dtype == np.complex128 and got.shape == (3, 3)
            assert np.array_equal(got.view(np.float64), want.view(np.float64)) or \
                np.array_equal(got, want), (th, dcp, reparam)
            assert np.array_equal(got, want)
    # sines set directly (osc_params.py:86-152), as Python floats and as numpy scalars
    for val in (0.3, np.float64(0.3)):
        o.sin12 = o.sin13 = o.sin23 = val
        assert np.array_equal(o.mix_matrix_complex, numpy_form(val, val, val, dcp, False))


def test_float_tuple_forms_of_the_matrices_are_the_matrices():
    """`OscParams.mix_floats / dm_floats` (what the fit loop writes straight into the kernels' parameter block) are,
    entry by entry and bit by bit, `mix_matrix_complex` / `mix_matrix_reparam_complex` / `dm_matrix`, degeneracy
    nudges included; `Prob3ParamsBlock.update` with them fills the block `make_prob3_params` builds."""
    from pisa_amd import _lib
    from pisa_amd.stages.osc.osc_params import OscParams

    rs = np.random.RandomState(0)
    blk = _lib.Prob3ParamsBlock()
    std = np.diag([1.0, 0, 0]).astype(complex)
    zc, zr = np.zeros((3, 3), complex), np.zeros((3, 3))
    for m in (std, zc, zr):
        m.setflags(write=False)
    for _ in range(500):
        o = OscParams()
        o.theta12, o.theta13, o.theta23 = rs.rand(3) * 1.5
        o.deltacp = rs.rand() * 2 * np.pi
        o.dm21 = rs.choice([0.0, 7.5e-5 * rs.rand()])
        o.dm31 = rs.choice([0.0, 2.5e-3 * (rs.rand() - 0.5)])
        for reparam, want in ((False, o.mix_matrix_complex), (True, o.mix_matrix_reparam_complex)):
            got = np.array(o.mix_floats(reparam)).view(np.complex128).reshape(3, 3)
            assert got.tobytes() == want.tobytes()
        assert np.array(o.dm_floats()).reshape(3, 3).tobytes() == o.dm_matrix.tobytes()
        a = blk.update(o.dm_floats(), o.mix_floats(), std, -1, zc, zr)
        b = _lib.make_prob3_params(o.dm_matrix, o.mix_matrix_complex, std, -1, zc, zr)
        assert bytes(a) == bytes(b)
