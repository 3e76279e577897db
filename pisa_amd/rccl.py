"""Direct RCCL all-reduce of the histogram limbs on the launch stream.

`torch.distributed.all_reduce` runs the collective on a stream of its own and brackets it with
two event hand-overs; in a fit loop whose whole evaluation is ~85 us that costs ~8 us per
evaluation.  This module binds the four RCCL entry points the limb all-reduce needs (ctypes on
the librccl.so that torch has already loaded -- one RCCL per process) and enqueues
`ncclAllReduce(int64, sum)` on the stream the kernels are launched on, between the fused kernel
and the tail kernel, with no stream hop.  The communicator is created once from a unique id that
rank 0 broadcasts through the existing torch.distributed group.  Anything that goes wrong at
set-up makes every rank fall back to `torch.distributed` (the ranks agree on that first).
"""
import ctypes as C
import os

NCCL_INT64 = 4  # ncclDataType_t
NCCL_SUM = 0    # ncclRedOp_t


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _load():
    import torch

    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    lib.ncclGetUniqueId.restype = C.c_int
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = C.c_int
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.restype = C.c_int
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]
    lib.ncclCommDestroy.restype = C.c_int
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclCommCount.restype = C.c_int
    lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib


class LimbAllReduce:
    """int64 SUM all-reduce over the ranks of a torch.distributed group, enqueued on a raw HIP
    stream.  `create` returns None when the direct path is not available on every rank."""

    def __init__(self, lib, comm, world_size):
        self.lib, self.comm, self.world_size = lib, comm, world_size

    @classmethod
    def create(cls, device, group=None):
        import torch
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
            return None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        lib, ok = None, 1
        uid = _UniqueId()
        try:
            lib = _load()
            if rank == 0:
                rc = lib.ncclGetUniqueId(C.byref(uid))
                if rc != 0:
                    raise RuntimeError(lib.ncclGetErrorString(rc).decode())
        except Exception:  # library or symbols missing: every rank falls back together
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            return None
        t = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8))
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        comm = C.c_void_p()
        rc = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
        flag.fill_(1 if rc == 0 else 0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if rc == 0:
                lib.ncclCommDestroy(comm)
            return None
        return cls(lib, comm, world)

    def all_reduce_(self, limbs, stream):
        """in-place SUM of an int64 device tensor over the ranks, on raw stream `stream`"""
        assert limbs.is_contiguous() and limbs.element_size() == 8
        rc = self.lib.ncclAllReduce(limbs.data_ptr(), limbs.data_ptr(), limbs.numel(), NCCL_INT64,
                                    NCCL_SUM, self.comm, stream)
        if rc != 0:
            raise RuntimeError("ncclAllReduce: " + self.lib.ncclGetErrorString(rc).decode())
        return limbs

    def count(self):
        """`ncclCommCount`: the number of ranks the communicator really spans"""
        n = C.c_int(-1)
        rc = self.lib.ncclCommCount(self.comm, C.byref(n))
        if rc != 0:
            raise RuntimeError("ncclCommCount: " + self.lib.ncclGetErrorString(rc).decode())
        return int(n.value)

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
