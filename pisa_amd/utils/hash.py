"""`hash_obj` / `hash_file` (pisa/utils/hash.py:66-181): an integer (or hex / base-64 text) that is the same for
objects that are equal after normalisation; md5 of a canonical byte form."""
import base64
import hashlib
import pickle
import struct
from collections.abc import Mapping

import numpy as np

__all__ = ["hash_obj", "hash_file"]


def _feed(h, obj):
    from pisa_amd.core.units import Quantity

    if hasattr(obj, "hashable_state") and not isinstance(obj, (Mapping, np.ndarray)):
        obj = obj.hashable_state
    elif hasattr(obj, "hash") and isinstance(getattr(obj, "hash"), int):
        obj = ("hash", obj.hash)
    if isinstance(obj, Quantity):
        h.update(b"Q")
        _feed(h, np.asarray(obj.magnitude) * obj.units.scale)
        h.update(repr(tuple(obj.units.dims)).encode())
    elif isinstance(obj, np.ndarray):
        h.update(b"A" + str(obj.dtype).encode() + repr(obj.shape).encode())
        h.update(np.ascontiguousarray(obj).tobytes() if obj.dtype != object else pickle.dumps(obj.tolist()))
    elif isinstance(obj, Mapping):
        h.update(b"M")
        for k in (obj.keys() if type(obj).__name__ == "OrderedDict" else sorted(obj.keys(), key=repr)):
            _feed(h, k)
            _feed(h, obj[k])
    elif isinstance(obj, (list, tuple)):
        h.update(b"S%d" % len(obj))
        for x in obj:
            _feed(h, x)
    elif isinstance(obj, (set, frozenset)):
        h.update(b"T")
        for x in sorted(obj, key=repr):
            _feed(h, x)
    elif isinstance(obj, (float, np.floating)):
        h.update(b"f" + struct.pack("<d", float(obj)))
    elif isinstance(obj, (bool, np.bool_)):
        h.update(b"b1" if obj else b"b0")
    elif isinstance(obj, (int, np.integer)):
        h.update(b"i" + str(int(obj)).encode())
    elif isinstance(obj, str):
        h.update(b"s" + obj.encode("utf-8"))
    elif isinstance(obj, bytes):
        h.update(b"y" + obj)
    elif obj is None:
        h.update(b"N")
    else:
        h.update(b"P" + pickle.dumps(obj, protocol=4))


def _as(digest, hash_to):
    if hash_to in (None, "i", "int", "integer"):
        return struct.unpack("<q", digest[:8])[0]
    if hash_to in ("b", "bin", "binary"):
        return digest
    if hash_to in ("h", "x", "hex", "hexadecimal"):
        return digest.hex()
    if hash_to in ("b64", "base64"):
        return base64.b64encode(digest).decode()
    raise ValueError('Unrecognized `hash_to`: "%s"' % (hash_to,))


def hash_obj(obj, hash_to="int", full_hash=True):
    """hash of `obj`: a signed 64-bit integer ('int'), hex text ('hex'), base-64 text ('base64') or the 16 bytes
    ('bin').  (`full_hash=False` -- the reference's sampling of large arrays -- hashes everything here as well.)"""
    h = hashlib.md5()
    _feed(h, obj)
    return _as(h.digest(), hash_to)


def hash_file(fname, hash_to=None, full_hash=True):
    with open(fname, "rb") as f:
        return _as(hashlib.md5(f.read()).digest(), hash_to)
