"""Reading and writing files by their extension: the interface of pisa/utils/fileio.py (`from_file` :486-539,
`to_file` :542-590, pickles, text arrays, cfg files, `expand`, `mkdir`, `nsort`).  JSON (also .bz2 / .xz) through
`utils/jsons.py`, HDF5 read through `utils/hdf.py` (writing HDF5 needs an HDF5 library and is not provided)."""
import os
import pickle
import re

import numpy as np

from pisa_amd.utils import jsons
from pisa_amd.utils.resources import find_resource

__all__ = ["JSON_EXTS", "HDF5_EXTS", "PKL_EXTS", "CFG_EXTS", "ZIP_EXTS", "TXT_EXTS", "expand", "mkdir", "nsort", "from_cfg",
           "from_pickle", "to_pickle", "from_txt", "to_txt", "from_file", "to_file"]

JSON_EXTS = ["json"]
HDF5_EXTS = ["hdf", "h5", "hdf5"]
PKL_EXTS = ["pickle", "pckl", "pkl", "p"]
CFG_EXTS = ["ini", "cfg"]
ZIP_EXTS = ["bz2", "xz"]
TXT_EXTS = ["txt", "dat"]


def expand(path, exp_user=True, exp_vars=True, absolute=False, resolve_symlinks=False):
    if exp_user:
        path = os.path.expanduser(path)
    if exp_vars:
        path = os.path.expandvars(path)
    if absolute:
        path = os.path.abspath(path)
    if resolve_symlinks:
        path = os.path.realpath(path)
    return path


def mkdir(d, mode=0o0750, warn=True):
    os.makedirs(d, mode=mode, exist_ok=True)


def nsort(l, reverse=False):  # noqa: E741 (the reference's argument name)
    """strings sorted so that the numbers in them count as numbers: f2 before f10"""
    return sorted(l, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)], reverse=reverse)


def _ext(fname):
    base, ext = os.path.splitext(str(fname))
    ext = ext.lstrip(".").lower()
    zipped = ext in ZIP_EXTS
    if zipped:
        ext = os.path.splitext(base)[1].lstrip(".").lower()
    return ext, zipped


def from_cfg(fname):
    from pisa_amd.core.config_parser import PISAConfigParser

    config = PISAConfigParser()
    config.read(find_resource(fname))
    return config


def from_pickle(fname):
    with open(find_resource(fname), "rb") as f:
        return pickle.load(f)


def to_pickle(obj, fname, overwrite=True, warn=True):
    if not overwrite and os.path.exists(fname):
        raise IOError("%s exists" % fname)
    with open(fname, "wb") as f:
        pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)


def from_txt(fname, as_array=False):
    if as_array:
        return np.loadtxt(find_resource(fname))
    with open(find_resource(fname), "r") as f:
        return f.read()


def to_txt(obj, fname):
    with open(fname, "w") as f:
        f.write(obj)


def from_file(fname, fmt=None, **kwargs):
    """the contents of a file of any of the known kinds, found through the resource path"""
    ext, _ = _ext(fname) if fmt is None else (fmt.lower(), False)
    if ext in JSON_EXTS:
        return jsons.from_json(find_resource(fname))
    if ext in HDF5_EXTS:
        from pisa_amd.utils.hdf import from_hdf

        return from_hdf(fname, **kwargs)
    if ext in PKL_EXTS:
        return from_pickle(fname)
    if ext in CFG_EXTS:
        return from_cfg(fname)
    if ext in TXT_EXTS:
        return from_txt(fname, **kwargs)
    raise TypeError("File %s has unrecognized extension: %s. Valid extensions are: %s"
                    % (fname, ext, JSON_EXTS + HDF5_EXTS + PKL_EXTS + CFG_EXTS + TXT_EXTS))


def to_file(obj, fname, fmt=None, overwrite=True, warn=True, **kwargs):
    ext, _ = _ext(fname) if fmt is None else (fmt.lower(), False)
    if not overwrite and os.path.exists(fname):
        raise IOError("%s exists" % fname)
    if ext in JSON_EXTS:
        return jsons.to_json(obj, fname, **kwargs)
    if ext in PKL_EXTS:
        return to_pickle(obj, fname, overwrite=overwrite, warn=warn)
    if ext in TXT_EXTS:
        if kwargs:
            raise ValueError("Following additional keyword arguments not accepted when writing to text file: %s" % list(kwargs))
        return to_txt(obj, fname)
    if ext in HDF5_EXTS:
        raise NotImplementedError("writing HDF5 needs an HDF5 library (h5py), which this build does not depend on;"
                                  " write .json / .pkl instead")
    raise TypeError("Unrecognized file type/extension: %s" % ext)
