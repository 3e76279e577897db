"""`Prior` under the module name the reference has it (pisa/core/prior.py) and `get_prior_bounds` (:372-438): the
parameter values at which a prior's chi2 crosses `stddev`^2, scanned on 10 000 points of its valid range."""
from collections import OrderedDict
from collections.abc import Iterable
from numbers import Number

import numpy as np

from pisa_amd.core.param import Prior
from pisa_amd.core.units import Quantity

__all__ = ["Prior", "get_prior_bounds"]


def get_prior_bounds(obj, param=None, stddev=1.0):
    """`obj`: a Prior, a dict holding one (optionally under 'params' / <param> / 'prior'), or a file of such a dict;
    returns {stddev: [values where chi2 crosses stddev**2, ...]}"""
    stddev = [stddev] if isinstance(stddev, Number) else list(stddev) if isinstance(stddev, Iterable) else [stddev]
    bounds = OrderedDict((s, []) for s in stddev)
    if isinstance(obj, Prior):
        prior = obj
    else:
        if isinstance(obj, str):
            from pisa_amd.utils.fileio import from_file

            obj = from_file(obj)
        if "params" in obj:
            obj = obj["params"]
        if param is not None and param in obj:
            obj = obj[param]
        if "prior" in obj:
            obj = obj["prior"]
        prior = Prior(**obj)
    x0, x1 = (q.magnitude for q in prior.valid_range)
    x = np.linspace(x0, x1, 10000)
    chi2 = prior.chi2(Quantity(x, prior.units))
    for i in range(len(x) - 1):
        for s in stddev:
            level = s ** 2
            if chi2[i] > level and chi2[i + 1] < level:
                bounds[s].append(Quantity(x[i], prior.units))
            elif chi2[i] < level and chi2[i + 1] > level:
                bounds[s].append(Quantity(x[i + 1], prior.units))
    return bounds
