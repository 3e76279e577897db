"""Ultrasurfaces: event-wise detector systematics from pre-fitted gradients (counterpart of
pisa/stages/discr_sys/ultrasurfaces.py:38-365).  Setup: every event takes the gradients `grad__<p>[__<q>...]` of its
nearest neighbour (in `varnames`, scikit-learn's KDTree as in the reference; optionally within its event grouping)
among the events of the feather file (a `.csv` with the same columns is read too).  Per parameter change one number per gradient -- the product of the
parameters' offsets from their nominal points, with the reference's three extrapolation rules beyond the `support`
-- and one pass over the gradient columns: `us_scales = exp(sum_g shift_g * grad_g)` (or `1 + sum`,
`approx_exponential`), `pisa_hip_column_combination`.  Per run `weights *= us_scales`."""
import collections.abc

import numpy as np

from pisa_amd import FTYPE
from pisa_amd import kernels as K
from pisa_amd.core.stage import Stage
from pisa_amd.utils.log import logging
from pisa_amd.utils.resources import find_resource

__all__ = ["get_us_grouping_from_container_name", "ultrasurfaces"]


def get_us_grouping_from_container_name(name, groupings_set):
    """the grouping (e.g. 'numu_numubar_cc', or the one ending in 'nc') the container `name` belonged to when the
    gradients were fitted (ultrasurfaces.py:38-82)"""
    assert len([group for group in groupings_set if group.lower().endswith("nc")]) == 1
    flav, int_type = name.lower().split("_")
    for group in groupings_set:
        if int_type == "cc" and ("%s_" % flav) in group.lower() and group.lower().endswith(int_type):
            return group
        if int_type == "nc" and group.lower().endswith(int_type):
            return group
    raise ValueError("Unable to find event grouping associated with %s among the groups %s!" % (name, groupings_set))


class ultrasurfaces(Stage):  # pylint: disable=invalid-name
    def __init__(self, fit_results_file, nominal_points, varnames, event_grouping_key=None, approx_exponential=False,
                 support=None, extrapolation="continue", distance_tol=1e-5, **std_kwargs):
        self.fit_results_file = find_resource(fit_results_file)
        self.varnames = list(varnames)
        assert isinstance(event_grouping_key, str) or event_grouping_key is None
        self.event_grouping_key = event_grouping_key
        self.approx_exponential = approx_exponential
        assert isinstance(distance_tol, (int, float))
        self.distance_tol = distance_tol
        self.nominal_points = eval(nominal_points) if isinstance(nominal_points, str) else nominal_points  # pylint: disable=eval-used
        assert isinstance(self.nominal_points, collections.abc.Mapping)
        if isinstance(support, str):
            self.support = eval(support)  # pylint: disable=eval-used
            assert isinstance(self.support, collections.abc.Mapping)
        elif isinstance(support, collections.abc.Mapping) or support is None:
            self.support = support
        else:
            raise ValueError("Unknown input format for `support`.")
        assert extrapolation in ["continue", "linear", "constant"]
        self.extrapolation = extrapolation
        param_names = list(self.nominal_points.keys())
        for pname in param_names:
            if self.support is not None and pname not in self.support:
                raise ValueError("Support range is missing for parameter %s" % pname)
        keys = self.varnames + ["weights"]
        if "true_energy" not in keys:
            keys.append("true_energy")
        super().__init__(expected_params=param_names, expected_container_keys=keys,
                         supported_reps={"calc_mode": "events"}, **std_kwargs)

    def setup_function(self):
        import pandas as pd
        from sklearn.neighbors import KDTree

        if self.fit_results_file.endswith(".csv"):
            # (beyond the reference, which reads feather only: the same table as text, for hosts without pyarrow)
            df = pd.read_csv(self.fit_results_file, float_precision="round_trip")
        else:
            df = pd.read_feather(self.fit_results_file)
        self.gradient_names = [key for key in df.keys() if key.startswith("grad")]
        points = df[self.varnames].to_numpy()
        if self.event_grouping_key is not None:
            groupings_array = df[self.event_grouping_key].to_numpy()
            groupings_set = set(groupings_array)
        else:
            tree = KDTree(points)
        for container in self.data:
            container["us_scales"] = np.ones(container.size, dtype=FTYPE)
            n_container = len(container["true_energy"])
            mine = np.zeros((n_container, len(self.varnames)), dtype=points.dtype)
            for i, vname in enumerate(self.varnames):
                mine[:, i] = container[vname]
            where = slice(None)
            if self.event_grouping_key is not None:
                where = np.where(groupings_array == get_us_grouping_from_container_name(container.name, groupings_set))
                tree = KDTree(points[where])
            dists, ind = tree.query(mine, k=1, return_distance=True, dualtree=False, breadth_first=False)
            n_outside_tol = int(np.sum(dists > self.distance_tol))
            if n_outside_tol:
                logging.warning("For %d %s events (%.2g%%), the nearest neighbor, from which each gradient will be taken,"
                                " is at a distance beyond the pre-set tolerance of %.2g. The maximum distance to a nearest"
                                " neighbor is %.2g.", n_outside_tol, container.name, n_outside_tol * 100.0 / n_container,
                                self.distance_tol, np.max(dists))
            for gradient_name in self.gradient_names:
                container[gradient_name] = np.ascontiguousarray(df[gradient_name].to_numpy()[where][ind.ravel()], dtype=FTYPE)

    def _shifts(self):
        """one factor per gradient column (ultrasurfaces.py:296-337)"""
        out = []
        for gradient_name in self.gradient_names:
            feature = 1.0
            param_names = gradient_name.split("grad")[-1].split("__")[1:]
            grad_order = len(param_names)
            has_interactions = len(set(param_names)) > 1
            for i, pname in enumerate(param_names):
                value = self.params[pname].m
                bounded = value if self.support is None else np.clip(value, *self.support[pname])
                x_b = bounded - self.nominal_points[pname]
                x = value - self.nominal_points[pname]
                if self.extrapolation == "continue":
                    feature *= x
                elif self.extrapolation == "constant":
                    feature *= x_b
                else:
                    if grad_order == 1:
                        feature *= x
                        continue
                    if has_interactions:
                        raise RuntimeError("Cannot calculate linear extrapolation for gradients with interaction terms: %s"
                                           % gradient_name)
                    if i == 0:
                        feature *= x_b
                    elif i == 1:
                        feature *= (2 * x - x_b)
                    else:
                        raise RuntimeError("Cannot use linear extrapolation for orders > 2")
            out.append(float(feature))
        return out

    def compute_function(self):
        shifts = self._shifts()
        for container in self.data:
            container["us_scales"] = K.column_combination([container.device(g) for g in self.gradient_names], shifts,
                                                          container.size, "one_plus" if self.approx_exponential else "exp")
            container.mark_valid("us_scales")

    def apply_function(self):
        for container in self.data:
            container["weights"] = K.bin_scale(container.device("weights"), container.device("us_scales"))


def init_test(**param_kwargs):
    """Instantiation example (what pisa_tests/test_services.py calls for every service; the reference's own values)"""
    import os
    import tempfile
    import warnings

    import pandas as pd

    from pisa_amd.core.param import Param, ParamSet

    p1, p2 = "opt_eff_overall", "ice_scattering"
    param_set = ParamSet([Param(name=p1, value=1.0, **param_kwargs), Param(name=p2, value=0.0, **param_kwargs)])
    nominal_points = {p1: param_set[p1].value.m_as("dimensionless"), p2: param_set[p2].value.m_as("dimensionless")}
    n = 100
    rs = np.random.RandomState(0)
    varnames = ["inelasticity", "reco_energy"]
    df = {var: rs.random_sample(n).astype(FTYPE) for var in varnames}
    df.update({"grad_%s" % p: np.multiply(rs.random_sample(n), 2).astype(FTYPE) for p in param_set.names})
    df["grad__%s__%s" % (p1, p2)] = np.multiply(rs.random_sample(n), 2).astype(FTYPE)
    path = os.path.join(tempfile.gettempdir(), "pisa_amd_test_us_file_%d.feather" % os.getpid())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pd.DataFrame.from_dict(data=df, dtype=FTYPE).to_feather(path)
        except ImportError:                       # no pyarrow on this host
            path = path.replace(".feather", ".csv")
            pd.DataFrame.from_dict(data=df, dtype=FTYPE).to_csv(path, index=False)
    return ultrasurfaces(params=param_set, fit_results_file=path, varnames=varnames, nominal_points=nominal_points,
                         calc_mode="events", apply_mode="events")
