"""The element-wise helpers third-party stages are written with (counterpart of pisa/utils/vectorizer.py:44-209):
`scale`, `mul`, `imul`, `imul_and_scale`, `itruediv` (0 where the divisor is 0), `assign`, `pow`, `sqrt`,
`replace_where_counts_gt`, all writing into `out`.  One launch of `pisa_hip_vector_op` each.  Arguments may be device
tensors (the natural form here: `container.device(key)`; `out` is then written in place -- follow with
`container.mark_dev_changed(key)` where the reference has `mark_changed`) or numpy arrays, which are uploaded, and
`out[...]` is filled from the device result."""
import numpy as np

from pisa_amd import FTYPE

__all__ = ["scale", "mul", "imul", "imul_and_scale", "itruediv", "assign", "pow", "sqrt", "replace_where_counts_gt"]


def _run(op, a, out, b=None, scalar=0.0):
    import torch

    from pisa_amd import kernels as K

    def dev(x):
        if x is None or isinstance(x, torch.Tensor):
            return x
        x = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=FTYPE), np.shape(out)))
        return K.to_device(x.ravel())

    if isinstance(out, torch.Tensor):
        K.vector_op(op, dev(a).reshape(-1), out.view(-1), None if b is None else dev(b).reshape(-1), scalar)
        return out
    res = K.to_device(np.ascontiguousarray(out, dtype=FTYPE).ravel())
    K.vector_op(op, dev(a).reshape(-1), res, None if b is None else dev(b).reshape(-1), scalar)
    out[...] = res.cpu().numpy().reshape(np.shape(out))
    return out


def scale(vals, scale, out):  # pylint: disable=redefined-outer-name
    """out[:] = vals[:] * scale"""
    return _run("scale", vals, out, scalar=scale)


def mul(vals0, vals1, out):
    """out[:] = vals0[:] * vals1[:]"""
    return _run("mul", vals0, out, b=vals1)


def imul(vals, out):
    """out[:] *= vals[:]"""
    return _run("imul", vals, out)


def imul_and_scale(vals, scale, out):  # pylint: disable=redefined-outer-name
    """out[:] *= vals[:] * scale"""
    return _run("imul_and_scale", vals, out, scalar=scale)


def itruediv(vals, out):
    """out[:] /= vals[:]; division by zero gives 0 for that element"""
    return _run("itruediv", vals, out)


def assign(vals, out):
    """out[:] = vals[:]"""
    return _run("assign", vals, out)


def pow(vals, pwr, out):  # pylint: disable=redefined-builtin
    """out[:] = vals[:] ** pwr"""
    return _run("pow", vals, out, scalar=pwr)


def sqrt(vals, out):
    """out[:] = sqrt(vals[:])"""
    return _run("sqrt", vals, out)


def replace_where_counts_gt(vals, counts, min_count, out):
    """out[i] = vals[i] where counts[i] > min_count"""
    return _run("replace_where_counts_gt", vals, out, b=counts, scalar=min_count)
