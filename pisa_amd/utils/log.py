"""Logging as the reference's modules use it (pisa/utils/log.py:47-143): `logging` with a TRACE level below DEBUG,
`Levels`, `set_verbosity`, and the `physics` / `tprofile` loggers."""
import enum
import logging

__all__ = ["Levels", "logging", "physics", "tprofile", "set_verbosity"]

_TRACE = 5
logging.addLevelName(_TRACE, "TRACE")
if not hasattr(logging, "trace"):
    logging.TRACE = _TRACE
    logging.trace = lambda msg, *args, **kwargs: logging.log(_TRACE, msg, *args, **kwargs)
    logging.Logger.trace = lambda self, msg, *args, **kwargs: self.log(_TRACE, msg, *args, **kwargs)


class Levels(enum.IntEnum):
    FATAL = 0
    ERROR = 0
    WARN = 0
    INFO = 1
    DEBUG = 2
    TRACE = 3


_PY_LEVEL = {0: logging.WARN, 1: logging.INFO, 2: logging.DEBUG, 3: _TRACE}
physics = logging.getLogger("physics")
tprofile = logging.getLogger("profile")


def set_verbosity(verbosity):
    """0 (warnings and worse) ... 3 (trace) for the root, physics and profile loggers"""
    level = _PY_LEVEL[min(max(int(verbosity), 0), 3)]
    if not logging.getLogger().handlers:
        logging.basicConfig(format="[%(levelname)8s] %(message)s")
    for lg in (logging.getLogger(), physics, tprofile):
        lg.setLevel(level)
