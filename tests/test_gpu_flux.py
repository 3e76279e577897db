"""flux.honda_ip on the device vs the reference's own values (golden vectors made by
importing pisa/utils/flux_weights.py, oracle/gen_golden.py:gen_flux) and vs the
oracle restatement on fresh points."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TABLE = "flux/honda-2015-spl-solmin-aa.d"
# spline interpolation of tabulated fluxes: the reference's per-event QR solve and
# our cardinal-spline product differ by rounding only
TOL = dict(rtol=1e-10, atol=0.0)


def test_flux_2d_matches_reference_goldens():
    from pisa_amd.utils import flux_weights as fw

    g = np.load(os.path.join(GOLD, "flux_ref.npz"))
    assert str(g["table"]) == TABLE
    table = fw.load_2d_table(TABLE)
    nu, nubar = fw.calculate_2d_flux_weights(g["true_energy"], g["true_coszen"], table)
    nu, nubar = nu.cpu().numpy(), nubar.cpu().numpy()
    np.testing.assert_allclose(nu[:, 0], g["nue"], **TOL)
    np.testing.assert_allclose(nu[:, 1], g["numu"], **TOL)
    np.testing.assert_allclose(nubar[:, 0], g["nuebar"], **TOL)
    np.testing.assert_allclose(nubar[:, 1], g["numubar"], **TOL)


def test_flux_2d_matches_oracle_and_rejects_bad_coszen():
    from oracle import flux_oracle
    from pisa_amd.utils import flux_weights as fw
    from pisa_amd.utils.resources import find_resource

    rs = np.random.RandomState(5)
    n = 3000
    e = 10 ** (rs.rand(n) * 4.5 - 1)
    cz = rs.rand(n) * 2 - 1
    table = fw.load_2d_table(TABLE)
    nu, nubar = fw.calculate_2d_flux_weights(e, cz, table)
    ref = flux_oracle.load_2d_honda_table(find_resource(TABLE))
    sel = slice(0, 300)  # the oracle follows the reference's per-event Python loop
    for col, prim in ((nu[:, 0], "nue"), (nu[:, 1], "numu"), (nubar[:, 0], "nuebar"), (nubar[:, 1], "numubar")):
        want = flux_oracle.calculate_2d_flux_weights(e[sel], cz[sel], ref[prim])
        np.testing.assert_allclose(col.cpu().numpy()[sel], want, **TOL)
    assert np.all(nu.cpu().numpy() > 0)
    cz[17] = 1.0000001
    with pytest.raises(ValueError):
        fw.calculate_2d_flux_weights(e, cz, table)
    with pytest.raises(ValueError):
        fw.load_2d_table("flux/honda-2015-spl-solmin.d")  # not azimuth averaged


def test_bartol_table_matches_reference_goldens():
    """`load_2d_table` on a Bartol table (flux_weights.py:133-203: Honda-like layout, two energy step
    widths) -> the band splines have non-uniform knots; same kernel.  Against values produced by the
    reference's own code (tests/golden/flux_bartol_ref.npz, oracle/gen_golden.py:gen_flux)."""
    from pisa_amd.utils import flux_weights as fw

    g = np.load(os.path.join(GOLD, "flux_bartol_ref.npz"))
    table = fw.load_2d_table(str(g["table"]))
    assert table["name"] == "bartol"
    nu, nubar = fw.calculate_2d_flux_weights(g["true_energy"], g["true_coszen"], table)
    nu, nubar = nu.cpu().numpy(), nubar.cpu().numpy()
    np.testing.assert_allclose(nu[:, 0], g["nue"], **TOL)
    np.testing.assert_allclose(nu[:, 1], g["numu"], **TOL)
    np.testing.assert_allclose(nubar[:, 0], g["nuebar"], **TOL)
    np.testing.assert_allclose(nubar[:, 1], g["numubar"], **TOL)
