"""A/B of two builds of the library (the in-tree one and scripts/dev/libpisa_hip_prev.so, built
from the previous commit) on the planned prob3 evaluation, interleaved in separate processes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1:
    from pisa_amd import _lib
    if sys.argv[1] != "cur":
        _lib.LIB_PATH = os.path.join(HERE, "libpisa_hip_%s.so" % sys.argv[1])
    sys.argv = sys.argv[:1]
    exec(open(os.path.join(HERE, "dev_probe8.py")).read())
else:
    for rep in range(2):
        for which in [w for w in ("prev", "nostore", "noamp", "cur") if w == "cur" or os.path.exists(os.path.join(HERE, "libpisa_hip_%s.so" % w))]:
            out = subprocess.run([sys.executable, __file__, which], capture_output=True, text=True).stdout
            print(which, out.strip().splitlines()[-1])
