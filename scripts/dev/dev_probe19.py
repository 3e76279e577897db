"""A/B of library builds (scripts/dev/libpisa_hip_<name>.so) on the event-mode benchmark."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from pisa_amd import _lib
    if sys.argv[2] != "cur":
        _lib.LIB_PATH = os.path.join(HERE, "libpisa_hip_%s.so" % sys.argv[2])
    sys.argv = [sys.argv[0]] + sys.argv[3:]
    exec(open(os.path.join(ROOT, "scripts", "bench_events.py")).read())
else:
    names = sorted(os.path.basename(f)[len("libpisa_hip_"):-3] for f in glob.glob(os.path.join(HERE, "libpisa_hip_*.so")))
    for rep in range(2):
        for which in names + ["cur"]:
            out = subprocess.run([sys.executable, __file__, "--child", which] + sys.argv[1:], capture_output=True, text=True)
            line = (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1]
            print(which, line[:230])
