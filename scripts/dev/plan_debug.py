import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, "30000", "5"])
os.environ["PISA_RESOURCES"] = tmp
os.environ["PISA_PLAN_DEBUG"] = "1"
from pisa_amd.core.distribution_maker import DistributionMaker
from pisa_amd.core.units import ureg
dm = DistributionMaker(["settings/pipeline/IceCube_3y_neutrinos.cfg", "settings/pipeline/IceCube_3y_muons.cfg"])
t = dm.get_outputs(return_sum=True)[0]
nu = dm.pipelines[0]
print("plan after first:", nu._plan)
dm.params["theta23"].value = 44.0 * ureg.degree
outs = [p.get_outputs() for p in dm.pipelines]
print([type(o).__name__ for o in outs], nu._plan)
tot = [o.total("t") for o in outs]
print([x._lazy is not None for x in tot])
s = tot[0] + tot[1]
print("sum lazy:", s._lazy is not None, getattr(s, "_extra", None) is not None)
t = dm.get_outputs(return_sum=True)[0]
print("return_sum lazy:", t._lazy is not None)
