"""`get_random_state` (pisa/utils/random_numbers.py:39-137): a `numpy.random.RandomState` with local state from
whatever describes one."""
from collections.abc import Sequence

import numpy as np

__all__ = ["get_random_state"]


def get_random_state(random_state, jumpahead=None):
    """None / 'rand' / 'random': fresh, unseeded; a RandomState: itself; an int: seeded; one to three ints: a seed
    composed of bit fields of 32 / 15+17 / 1+12+19 bits; a 5-tuple: a state for `set_state`"""
    if jumpahead is not None:
        raise DeprecationWarning("`jumpahead` is deprecated since it does not result in an independent random"
                                 " sequence, simply use a different seed")
    if random_state is None:
        return np.random.RandomState()
    if isinstance(random_state, np.random.RandomState):
        return random_state
    if isinstance(random_state, str):
        if random_state.lower().strip() not in ("rand", "random"):
            raise ValueError("`random_state`=%s not a valid string. Must be one of %s." % (random_state, ["rand", "random"]))
        return np.random.RandomState()
    if isinstance(random_state, (int, np.integer)) and not isinstance(random_state, bool):
        return np.random.RandomState(seed=int(random_state))
    if isinstance(random_state, Sequence):
        rs = np.random.RandomState()
        if all(isinstance(x, (int, np.integer)) for x in random_state):
            fields = {1: (32,), 2: (15, 17), 3: (1, 12, 19)}.get(len(random_state))
            if fields is None:
                raise ValueError("`random_state` sequence of int must be length 1-3")
            seed = 0
            for x, bits in zip(random_state, fields):
                assert 0 <= x < 2 ** bits
                seed = (seed << bits) + int(x)
            rs.seed(seed)
        elif len(random_state) == 5:
            rs.set_state(random_state)
        else:
            raise ValueError("Do not know what to do with `random_state` Sequence %s" % (random_state,))
        return rs
    raise TypeError("Unhandled `random_state` of type %s: %s" % (type(random_state), random_state))
