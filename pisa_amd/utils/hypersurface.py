"""Hypersurface fits of discrete-systematics sets (the part of
pisa/utils/hypersurface/hypersurface.py that the published IceCube 3-year
analysis chain uses): the CSV files of the public data release
(`_load_hypersurfaces_data_release`, :2065-2173) with linear terms
(`linear_hypersurface_func`, :81-100) and `Hypersurface.evaluate` (:356-461) for
them.  Fit files written by `fit_hypersurfaces` (json), interpolated
hypersurfaces and uncertainty propagation (needs the fit covariance, which the
data release does not contain) are not part of this build.
"""
from collections import OrderedDict

import numpy as np
import pandas as pd

from pisa_amd.utils.resources import find_resource

__all__ = ["Hypersurface", "load_hypersurfaces"]


class Hypersurface:
    """scale[bin] = intercept[bin] + sum_p gradient_p[bin] * value_p (raw parameter
    values: `using_legacy_data`, hypersurface.py:432)."""

    def __init__(self, binning, param_names, intercept, gradients):
        self.binning = binning
        self.param_names = list(param_names)
        self.intercept = np.asarray(intercept, dtype=np.float64).reshape(binning.shape)
        self.gradients = OrderedDict(
            (n, np.asarray(g, dtype=np.float64).reshape(binning.shape)) for n, g in zip(param_names, gradients))
        self.using_legacy_data = True

    def evaluate(self, param_values, return_uncertainty=False):
        if return_uncertainty:
            raise NotImplementedError("the data-release hyperplanes carry no fit covariance")
        out = np.array(self.intercept, dtype=np.float64)
        for name in self.param_names:  # same accumulation order as :430-433
            value = param_values[name]
            assert np.isscalar(value), "sys param values must be a scalar when evaluating all bins simultaneously"
            out += self.gradients[name] * value
        return out


def load_hypersurfaces(input_file, expected_binning=None):
    """{map name: Hypersurface}; `input_file` = '<dir>/hyperplanes_*.csv[.bz2]'"""
    assert isinstance(input_file, str)
    if not (input_file.endswith("csv") or input_file.endswith("csv.bz2")):
        raise NotImplementedError("only the data-release CSV hyperplanes are part of this build")
    assert expected_binning is not None, "Must provide binning when loading data release hypersurfaces"
    binning = expected_binning
    files = OrderedDict([("nue_cc+nuebar_cc", "nue_cc"), ("numu_cc+numubar_cc", "numu_cc"),
                         ("nutau_cc+nutaubar_cc", "nutau_cc"), ("nu_nc+nubar_nc", "all_nc")])
    out = OrderedDict()
    param_names = None
    for map_name, tag in files.items():
        table = pd.read_csv(find_resource(input_file.replace("*", tag)))
        for n in binning.names:
            midpoints = np.unique(table.pop(n).values)
            assert midpoints.size == binning[n].num_bins, \
                "Mismatch between expected and actual binning dimensions"
        offset = table.pop("offset")
        if param_names is None:
            param_names = table.columns.tolist()
        else:
            assert param_names == table.columns.tolist(), \
                "Mismatch between hypersurface params in different files"
        out[map_name] = Hypersurface(binning, param_names, offset.values,
                                     [table[n].values for n in param_names])
    return out
