"""pisa_amd -- MI355X-native implementation of PISA's per-event hot path
(prob3 oscillation -> reweight -> weighted histogram -> LLH) behind PISA's
Stage / Container / Pipeline interface.  See DESIGN.md.

FTYPE is fixed to float64 (PISA's default, pisa/__init__.py:152-179); the only
compute target is the HIP library (there is no CPU fallback).
"""
import numpy as np

FTYPE = np.float64
CTYPE = np.complex128
ITYPE = np.int64
TARGET = "hip"
HASH_SIGFIGS = 12      # significant figures kept when values are normalised for hashing (pisa/__init__.py:277)

__version__ = "0.1.0"
