"""Development probe: read-only streaming ceilings on this box (see stream_probe.hip)."""
import ctypes as C
import os

import torch

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_probe.so"))
lib.probe.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
nbytes = 400_000_000
x = torch.zeros(nbytes // 8, dtype=torch.float64, device="cuda")
out = torch.zeros(1, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
names = {0: "grid-stride x1", 1: "grid-stride x2", 2: "grid-stride x4", 3: "grid-stride x8", 4: "grid-stride x4 nontemporal",
         5: "block chunks x2", 6: "block chunks x4", 7: "block chunks x8",
         8: "3 columns, prefetch", 9: "3 columns, plain", 10: "3 columns, 2 sweeps", 11: "3 columns, 4 sweeps"}
for variant in (0, 4, 8, 9, 10, 11):
    for blocks, threads in ((512, 1024), (768, 1024), (1024, 1024), (2048, 512), (2048, 256), (4096, 256)):
        ms = C.c_float()
        rc = lib.probe(x.data_ptr(), nbytes, out.data_ptr(), variant, blocks, threads, 20, C.byref(ms))
        print("%-28s %5d x %4d: %6.1f us  %.2f TB/s  rc=%d" % (names[variant], blocks, threads, ms.value * 1e3,
                                                               nbytes / ms.value / 1e9, rc), flush=True)
