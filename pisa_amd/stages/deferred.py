"""Deferred operations on a container's `weights`.

The reference applies every stage's `apply_function` immediately on host numpy
arrays: `weights = copy(initial_weights)` (loader), `weights *= flux.prob`
(prob3.py:621-622), `weights *= weighted_aeff*scale` (aeff.py:87), then three
histogram passes (hist.py:198-209).  Here the stages RECORD these operations on
the container; `utils.hist` recognises the chain [reset, osc, aeff] and runs it
as ONE pass over HBM (`pisa_hip_reweight_hist`).  Any other access to
`container['weights']` first materialises the pending chain with the unfused
kernels (`pisa_hip_apply_osc_weights`, `pisa_hip_apply_aeff`), so third-party
stages always observe the values the reference would have produced.
"""
from pisa_amd import kernels as K

KEY = "weights"


def _ops(container):
    return container.pending.setdefault(KEY, [])


def reset_weights(container):
    """weights = copy(initial_weights)  (toy_event_generator.py:101-104); recorded in event AND in map
    representations (a binned pipeline such as osc_example.cfg: `ContainerSet.get_mapset` runs the chains of
    all containers in one launch, `pisa_hip_weight_chain_multi`)"""
    container.pending[KEY] = [("reset",)]
    container.touch_pending(KEY)


def chain_open(container):
    """a chain that started with `reset` in the container's CURRENT representation is pending: later steps can
    be appended to it (anything else -- weights set some other way, or in another representation -- is
    applied at once by the caller, through `container.device('weights')` and its automatic translation)"""
    ops = container.pending.get(KEY)
    return bool(ops) and ops[0][0] == "reset" and container._pending_hash.get(KEY) == container._rep_hash


def _append(container, op):
    ops = container.pending.get(KEY)
    if ops and container._pending_hash.get(KEY) == container._rep_hash:
        # a chain is already pending here: the first step has moved the change counters and the validity bits
        # (`touch_pending`), and nobody has read the weights since (a read materialises and removes the chain)
        ops.append(op)
        return
    _ops(container).append(op)
    container.touch_pending(KEY)


def osc(container, flux_key="nu_flux"):
    """weights *= flux[:,0]*prob_e + flux[:,1]*prob_mu  (prob3.py:621-622)"""
    _append(container, ("osc", flux_key))


def aeff(container, scale):
    """weights *= weighted_aeff * scale  (aeff.py:87)"""
    _append(container, ("aeff", float(scale)))


def materialize(container, key=KEY):
    """Apply the pending chain with the one-stage-at-a-time kernels."""
    ops = container.pending.pop(key, [])
    if not ops:
        return
    for op in ops:
        if op[0] == "reset":
            container._store(key, container.device("initial_weights").clone())
        elif op[0] == "osc":
            w = container.device_raw(key)
            K.apply_osc_weights(container.device(op[1]), container.device("prob_e"),
                                container.device("prob_mu"), w)
            container._store(key, w)
        elif op[0] == "aeff":
            w = container.device_raw(key)
            K.apply_aeff(container.device("weighted_aeff"), op[1], w)
            container._store(key, w)


def batch_chain(container):
    """(flux_key or None, aeff scale or None) if the pending chain is reset [-> osc] [-> aeff] in the current
    representation: the shape `pisa_hip_weight_chain_multi` runs for many containers at once"""
    if not chain_open(container):
        return None
    ops = container.pending[KEY][1:]
    flux = scale = None
    if ops and ops[0][0] == "osc":
        flux, ops = ops[0][1], ops[1:]
    if ops and ops[0][0] == "aeff":
        scale, ops = ops[0][1], ops[1:]
    return None if ops else (flux, scale)


def fusable_chain(container):
    """(flux_key, scale) if the pending chain is exactly reset -> osc -> aeff."""
    ops = container.pending.get(KEY, [])
    if len(ops) == 3 and ops[0][0] == "reset" and ops[1][0] == "osc" and ops[2][0] == "aeff":
        return ops[1][1], ops[2][1]
    return None
