"""ctypes binding of ``libpisa_hip.so`` (C ABI: ``include/pisa_hip.h``).

The shared library is the product; this module only marshals arguments.
There is NO CPU fallback: if the library is missing, fails to load, or a call
returns a non-zero status, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PISA_HIP_LIB: development override (A/B of two builds of the library, scripts/dev)
LIB_PATH = os.environ.get("PISA_HIP_LIB") or os.path.join(_HERE, "libpisa_hip.so")

MAX_SHELLS = 64
MAX_DIMS = 3
ACC_LIMBS = 6
MAX_LAYERS = 120
MAX_POINTS = 16

ERR_NEGATIVE = -5
ERR_OVERFLOW = -6


class PisaHipError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("libpisa_hip: %s (status %d)" % (text, status))
        self.status = status


class Prob3Params(C.Structure):
    _fields_ = [
        ("dm", C.c_double * 9),
        ("mix", C.c_double * 18),
        ("mat_pot", C.c_double * 18),
        ("mat_decay", C.c_double * 18),
        ("lri_pot", C.c_double * 9),
        ("decay_flag", C.c_int64),
    ]


class Earth(C.Structure):
    _fields_ = [
        ("n_shell", C.c_int32),
        ("r_detector", C.c_double),
        ("radii", C.c_double * MAX_SHELLS),
        ("rhos", C.c_double * MAX_SHELLS),
        ("coszen_limit", C.c_double * MAX_SHELLS),
    ]


class Binning(C.Structure):
    _fields_ = [
        ("ndim", C.c_int32),
        ("nbins", C.c_int64 * MAX_DIMS),
        ("mins", C.c_double * MAX_DIMS),
        ("maxs", C.c_double * MAX_DIMS),
    ]


class EventSet(C.Structure):
    _fields_ = [
        ("n_events", C.c_int64),
        ("d_energy", C.c_void_p),
        ("d_coszen", C.c_void_p),
        ("d_probability", C.c_void_p),
        ("d_pepmu", C.c_void_p),
        ("nubar", C.c_int32),
        ("flav", C.c_int32),
    ]


class Container(C.Structure):
    _fields_ = [
        ("n_events", C.c_int64),
        ("d_grid_x", C.c_void_p),
        ("d_grid_y", C.c_void_p),
        ("d_nu_flux", C.c_void_p),
        ("d_weighted_aeff", C.c_void_p),
        ("d_initial_weights", C.c_void_p),
        ("d_sample", C.c_void_p * MAX_DIMS),
        ("d_node", C.c_void_p),
        ("d_bin", C.c_void_p),
        ("d_node_bin", C.c_void_p),
        ("d_aeff_w0", C.c_void_p),
        ("d_pepmu", C.c_void_p),
        ("flav", C.c_int32),
        ("nubar", C.c_int32),
        ("scale", C.c_double),
        ("d_weighted_flux", C.c_void_p),
        ("d_node_bin16", C.c_void_p),
        ("d_weighted_flux_q", C.c_void_p),
        ("d_part_start", C.c_void_p),
        ("n_part", C.c_int32),
        ("part_width", C.c_int32),
    ]


class FoldSet(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("d_flux", C.c_void_p),
        ("d_perm", C.c_void_p),
        ("d_static_w", C.c_void_p),
        ("d_out", C.c_void_p),
        ("layout", C.c_int32),
        ("reserved", C.c_int32),
    ]


class BarrSet(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("d_true_energy", C.c_void_p),
        ("d_true_coszen", C.c_void_p),
        ("d_nu_flux_nominal", C.c_void_p),
        ("d_nubar_flux_nominal", C.c_void_p),
        ("d_out", C.c_void_p),
        ("nubar", C.c_int32),
        ("reserved", C.c_int32),
    ]


class BarrFoldSet(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("d_nu_flux_nominal", C.c_void_p),
        ("d_nubar_flux_nominal", C.c_void_p),
        ("d_factors", C.c_void_p),
        ("d_static_w", C.c_void_p),
        ("d_out", C.c_void_p),
        ("nubar", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PackSet(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("n_pad", C.c_int64),
        ("d_perm", C.c_void_p),
        ("d_grid_x", C.c_void_p),
        ("d_grid_y", C.c_void_p),
        ("d_nu_flux", C.c_void_p),
        ("d_weighted_aeff", C.c_void_p),
        ("d_initial_weights", C.c_void_p),
        ("d_sample", C.c_void_p * 3),
        ("d_node", C.c_void_p),
        ("d_bin", C.c_void_p),
        ("o_grid_x", C.c_void_p),
        ("o_grid_y", C.c_void_p),
        ("o_nu_flux", C.c_void_p),
        ("o_weighted_aeff", C.c_void_p),
        ("o_initial_weights", C.c_void_p),
        ("o_sample", C.c_void_p * 3),
        ("o_node", C.c_void_p),
        ("o_bin", C.c_void_p),
        ("o_node_bin", C.c_void_p),
        ("o_aeff_w0", C.c_void_p),
        ("o_static_w", C.c_void_p),
        ("o_node_bin16", C.c_void_p),
        ("n_sample", C.c_int32),
        ("reserved", C.c_int32),
    ]


class KdeJob(C.Structure):
    _fields_ = [
        ("d_x", C.c_void_p),
        ("d_weights", C.c_void_p),
        ("d_index", C.c_void_p),
        ("n", C.c_int64),
        ("d_out", C.c_void_p),
        ("sum_w", C.c_double),
        ("pairs_pilot", C.c_int64),
        ("pairs_eval", C.c_int64),
        ("status", C.c_int32),
        ("reserved", C.c_int32),
    ]


class FluxTable(C.Structure):
    _fields_ = [
        ("n_bands", C.c_int32),
        ("n_knots_e", C.c_int32),
        ("d_knots_e", C.c_void_p),
        ("d_coef_e", C.c_void_p),
        ("n_knots_cz", C.c_int32),
        ("enpow", C.c_int32),
        ("d_knots_cz", C.c_void_p),
        ("d_cardinal", C.c_void_p),
        ("cz_step", C.c_double),
    ]


class ChainSet(C.Structure):
    """pisa_hip_chain_set"""
    _fields_ = [
        ("n", C.c_int64),
        ("d_initial_weights", C.c_void_p),
        ("d_nu_flux", C.c_void_p),
        ("d_prob_e", C.c_void_p),
        ("d_prob_mu", C.c_void_p),
        ("prob_stride", C.c_int64),
        ("d_weighted_aeff", C.c_void_p),
        ("aeff_scale", C.c_double),
        ("d_weights", C.c_void_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)


class EvaluatorDesc(C.Structure):
    """pisa_hip_evaluator_desc"""
    _fields_ = [
        ("h_containers", C.c_void_p),
        ("n_containers", C.c_int32),
        ("n_e", C.c_int32),
        ("e_major", C.c_int32),
        ("reserved", C.c_int32),
        ("h_calc_grid", C.c_void_p),
        ("h_out_binning", C.c_void_p),
        ("plan", C.c_void_p),
        ("d_energy", C.c_void_p),
        ("d_pepmu", C.c_void_p),
        ("d_limbs", C.c_void_p),
        ("d_hist", C.c_void_p),
        ("d_sumw2", C.c_void_p),
        ("partial", C.c_void_p),
        ("d_status", C.c_void_p),
        ("d_metric_status", C.c_void_p),
        ("allreduce", C.c_void_p),
        ("comm", C.c_void_p),
    ]


class KdeInfo(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("cells", C.c_int32 * 3),
        ("n_src", C.c_int64), ("n_cells", C.c_int64),
        ("factor", C.c_double), ("norm", C.c_double), ("sum_w", C.c_double),
        ("mean", C.c_double * 3), ("covariance", C.c_double * 9), ("inv_cov", C.c_double * 9),
        ("r_cut", C.c_double), ("cell", C.c_double),
        ("pairs_pilot", C.c_int64), ("pairs_eval", C.c_int64), ("n_dense", C.c_int32),
    ]


_SIGS = {
    "pisa_hip_strerror": (C.c_char_p, [C.c_int]),
    "pisa_hip_last_hip_error": (C.c_char_p, []),
    "pisa_hip_version": (C.c_int, []),
    "pisa_hip_device_count": (C.c_int, []),
    "pisa_hip_propagate_array": (C.c_int, [C.POINTER(Prob3Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_propagate_array_host": (C.c_int, [C.POINTER(Prob3Params), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "pisa_hip_prob3_grid": (C.c_int, [C.POINTER(Prob3Params), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_grid_plan_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "pisa_hip_grid_plan_destroy": (C.c_int, [C.c_void_p]),
    "pisa_hip_prob3_grid_planned": (C.c_int, [C.POINTER(Prob3Params), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_calc_layers": (C.c_int, [C.POINTER(Earth), C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_prob3_events": (C.c_int, [C.POINTER(Prob3Params), C.POINTER(Earth), C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_prob3_events_multi": (C.c_int, [C.POINTER(Prob3Params), C.POINTER(Earth), C.POINTER(EventSet), C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_fill_probs": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_lookup_regular": (C.c_int, [C.POINTER(Binning), C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_histogram_regular": (C.c_int, [C.POINTER(Binning), C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_event_indices": (C.c_int, [C.POINTER(Binning), C.POINTER(C.c_void_p), C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_reweight_hist": (C.c_int, [C.POINTER(Container), C.c_int32, C.POINTER(Binning), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Binning), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_reweight_hist_acc": (C.c_int, [C.POINTER(Container), C.c_int32, C.POINTER(Binning), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Binning), C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_finalize_metric": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_finalize_metric_scaled": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_prob3_grid_planned_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_reweight_hist_multi": (C.c_int, [C.POINTER(Container), C.c_int32, C.POINTER(Binning), C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(Binning), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_multi_points_per_pass": (C.c_int, [C.c_int64]),
    "pisa_hip_finalize_metric_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_finalize_metric_split": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_finalize_metric_parts": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_profile_events": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pisa_hip_hist_window_bins": (C.c_int, [C.c_int64]),
    "pisa_hip_hist_workgroups": (C.c_int, [C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32)]),
    "pisa_hip_deposit_block_order_workspace": (C.c_int64, [C.c_int64]),
    "pisa_hip_deposit_block_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pisa_hip_partition_order_workspace": (C.c_int64, [C.c_int64]),
    "pisa_hip_partition_order_sort": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.c_void_p]),
    "pisa_hip_partition_order_assemble": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                    C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pisa_hip_pack_resident_columns": (C.c_int, [C.POINTER(PackSet), C.c_void_p]),
    "pisa_hip_weight_chain_multi": (C.c_int, [C.POINTER(ChainSet), C.c_int32, C.c_void_p]),
    "pisa_hip_evaluator_create": (C.c_int, [C.POINTER(EvaluatorDesc), C.POINTER(C.c_void_p)]),
    "pisa_hip_evaluator_destroy": (C.c_int, [C.c_void_p]),
    "pisa_hip_evaluator_set_scale": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "pisa_hip_evaluator_eval": (C.c_int, [C.c_void_p, C.POINTER(Prob3Params), C.c_int32, C.c_void_p, C.c_int32, C.c_int64,
                                          C.POINTER(C.c_double), C.c_void_p]),
    "pisa_hip_apply_osc_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_apply_osc_weights_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_apply_aeff": (C.c_int, [C.c_void_p, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_hist_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_kde_eval": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_kde_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int64]),
    "pisa_hip_kde_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.c_void_p]),
    "pisa_hip_kde_resident_bytes": (C.c_int64, [C.c_void_p]),
    "pisa_hip_kde_eval_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "pisa_hip_kde_lattice_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_kde_evaluate_lattice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_kde_evaluate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_kde_info": (C.c_int, [C.c_void_p, C.POINTER(KdeInfo)]),
    "pisa_hip_kde_lattice_batch": (C.c_int, [C.POINTER(KdeJob), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_kde_lattice_submit": (C.c_int, [C.POINTER(KdeJob), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_kde_lattice_wait": (C.c_int, []),
    "pisa_hip_kde_lattice_wait_jobs": (C.c_int, [C.c_void_p, C.c_int32]),
    "pisa_hip_kde_pool_release": (C.c_int, []),
    "pisa_hip_kde_arrays": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pisa_hip_kde_destroy": (C.c_int, [C.c_void_p]),
    "pisa_hip_kde_configure": (C.c_int, [C.c_int32]),
    "pisa_hip_kde_release_scratch": (C.c_int, []),
    "pisa_hip_metric": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_bin_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_bin_sqrt": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_lookup_indices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_two_nu_osc": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_power_law": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_shift_toward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_poly_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_column_combination": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_vector_op": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_interp_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_decoherence_probs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_flux_2d": (C.c_int, [C.POINTER(FluxTable), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_transform_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_flux_prob_tables": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_fold_flux": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "pisa_hip_barr_simple": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_void_p, C.c_void_p]),
    "pisa_hip_fold_flux_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "pisa_hip_barr_simple_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "pisa_hip_barr_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pisa_hip_barr_fold_multi": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "pisa_hip_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    "pisa_hip_free": (C.c_int, [C.c_void_p]),
    "pisa_hip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pisa_hip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pisa_hip_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "pisa_hip_stream_synchronize": (C.c_int, [C.c_void_p]),
    "pisa_hip_set_device": (C.c_int, [C.c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib = None


def lib():
    """Load libpisa_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pisa_amd/csrc`.  There is no CPU fallback." % LIB_PATH
        )
    try:
        import torch  # noqa: F401  (loads libamdhip64.so first -> single runtime instance)
    except Exception:  # pragma: no cover - torch is optional for the C ABI itself
        pass
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(status):
    if status != 0:
        l = lib()
        text = l.pisa_hip_strerror(int(status)).decode()
        if status == -3:
            text += " -- " + l.pisa_hip_last_hip_error().decode()
        if status == ERR_NEGATIVE:
            raise ValueError(text)  # stats.py:231-240 raises ValueError
        if status == ERR_OVERFLOW:
            raise OverflowError(text)
        raise PisaHipError(status, text)


class Prob3ParamsBlock:
    """ONE parameter block rewritten in place point after point (a fit loop's serial path: the library copies the
    block during the call).  A matrix that is the very READ-ONLY ndarray handed over last time is not converted again
    (anything that could have been edited in place -- a writeable array, a list, a wrapper object -- is converted at
    every call).  `update` returns a VIEW of the one shared buffer: a caller that keeps several points needs
    `Prob3Params.from_buffer_copy(block.update(...))` (as `FastPlan.metric_many` does)."""

    def __init__(self):
        self.buf = np.zeros(73, np.float64)
        self.params = Prob3Params.from_buffer(self.buf)
        self._last = [None] * 5
        self._flag = None

    def update(self, dm, mix, mat_pot, decay_flag, mat_decay, lri_pot):
        """`dm` / `mix`: the arrays, or their 9 / 18 entries as plain floats (OscParams.dm_floats / mix_floats)"""
        buf, last = self.buf, self._last
        buf[0:9] = dm if type(dm) is tuple else dm.reshape(9)
        buf[9:27] = mix if type(mix) is tuple else mix.reshape(9).view(np.float64)
        if mat_pot is not last[2]:
            buf[27:45] = np.ascontiguousarray(mat_pot, np.complex128).reshape(9).view(np.float64)
            last[2] = mat_pot if isinstance(mat_pot, np.ndarray) and not mat_pot.flags.writeable else None
        if mat_decay is not last[3]:
            buf[45:63] = np.ascontiguousarray(mat_decay, np.complex128).reshape(9).view(np.float64)
            last[3] = mat_decay if isinstance(mat_decay, np.ndarray) and not mat_decay.flags.writeable else None
        if lri_pot is not last[4]:
            buf[63:72] = np.asarray(lri_pot, np.float64).reshape(9)
            last[4] = lri_pot if isinstance(lri_pot, np.ndarray) and not lri_pot.flags.writeable else None
        if decay_flag != self._flag:
            buf[72:73].view(np.int64)[0] = int(decay_flag)
            self._flag = decay_flag
        return self.params


def make_prob3_params(dm, mix, mat_pot, decay_flag, mat_decay, lri_pot):
    """the parameter block, assembled in one numpy buffer (slice assignment into the ctypes fields
    converts element by element and costs 9 us per block; this is 3 us)"""
    buf = np.empty(73, np.float64)
    buf[0:9] = np.asarray(dm, np.float64).reshape(9)
    buf[9:27] = np.ascontiguousarray(mix, np.complex128).reshape(9).view(np.float64)
    buf[27:45] = np.ascontiguousarray(mat_pot, np.complex128).reshape(9).view(np.float64)
    buf[45:63] = np.ascontiguousarray(mat_decay, np.complex128).reshape(9).view(np.float64)
    buf[63:72] = np.asarray(lri_pot, np.float64).reshape(9)
    buf[72:73].view(np.int64)[0] = int(decay_flag)
    return Prob3Params.from_buffer(buf)


def make_earth(radii, rhos, coszen_limit, r_detector):
    n = len(radii)
    if n > MAX_SHELLS:
        raise ValueError("Earth model has %d shells; at most %d supported" % (n, MAX_SHELLS))
    e = Earth()
    e.n_shell = n
    e.r_detector = float(r_detector)
    for k in range(n):
        e.radii[k] = float(radii[k])
        e.rhos[k] = float(rhos[k])
        e.coszen_limit[k] = float(coszen_limit[k])
    return e


def make_binning(mins, maxs, nbins):
    b = Binning()
    b.ndim = len(nbins)
    if not 1 <= b.ndim <= MAX_DIMS:
        raise ValueError("can only do up to 3D at the moment")  # translation.py:252
    for k in range(b.ndim):
        b.nbins[k] = int(nbins[k])
        b.mins[k] = float(mins[k])
        b.maxs[k] = float(maxs[k])
    return b
