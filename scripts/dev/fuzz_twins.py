"""Random walk of a DistributionMaker through the evaluation plan against a twin that takes the Stage protocol, and
against a freshly built maker every so often (development tool, GPU box).  No oracle here: the walk hunts STATE defects --
a stale memo, a replay that missed a change, a cache keyed on too little -- in pipelines the oracle-backed walk
(`fuzz_pipeline.py`) does not cover: the published 3-year analysis (csv_loader -> honda_ip -> barr_simple -> prob3 ->
aeff -> hist -> hypersurfaces, plus the muon template; a synthetic MC file stands in for the release's), and the binned
`osc_example.cfg`.  Every step a random subset of the makers' parameters (free AND fixed ones with a range) moves, sometimes
the ordering selection is switched, parameters are fixed / freed, `reset_free` is called, a point is evaluated twice; the
walkers' total templates (values and errors) must agree bit for bit... to 1e-12, the fresh maker's to 1e-12 as well.
usage: fuzz_twins.py {3y|osc} [steps] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, ".")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else "3y"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rs = np.random.RandomState(seed)
if which == "3y":
    tmp = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synthetic_3y_mc.py"), tmp, "24000", str(seed)])
    os.environ["PISA_RESOURCES"] = tmp
    CFGS = ["settings/pipeline/IceCube_3y_neutrinos.cfg", "settings/pipeline/IceCube_3y_muons.cfg"]
else:
    CFGS = ["settings/pipeline/osc_example.cfg"]

from pisa_amd.core.distribution_maker import DistributionMaker  # noqa: E402


def build(fast):
    m = DistributionMaker(list(CFGS))
    for p in m.pipelines:
        p.fast_path = fast
    return m


fast, slow = build(True), build(False)
SKIP = {"earth_model", "detector_depth", "prop_height", "n_events", "seed", "random", "livetime"} if which == "osc" else \
    {"earth_model", "detector_depth", "prop_height"}
names = [p.name for p in fast.params if p.range is not None and p.name not in SKIP and not isinstance(p.value, str)]


def template(maker):
    out = maker.get_outputs(return_sum=True)
    m = out[0]
    return np.array(m.hist, copy=True), np.array(m.std_devs, copy=True)


def same(a, b):
    return np.allclose(a[0], b[0], rtol=1e-12, atol=1e-13 * max(np.abs(b[0]).max(), 1e-300)) and \
        np.allclose(a[1], b[1], rtol=1e-12, atol=1e-13 * max(np.abs(b[1]).max(), 1e-300))


bad = 0
moved = 0           # steps after which the template differs from the step before (the walk is not vacuous)
last = None
t0 = time.time()
log = []
for step in range(steps):
    action = []
    r = rs.rand()
    if r < 0.07 and fast.param_selections:
        new = "ih" if "nh" in fast.param_selections else "nh"
        for m in (fast, slow):
            m.select_params(new)
        action.append("select " + new)
    elif r < 0.11:
        for m in (fast, slow):
            m.reset_free()
        action.append("reset_free")
    elif r < 0.15:
        n = names[rs.randint(len(names))]
        flag = not fast.params[n].is_fixed
        for m in (fast, slow):
            m.params[n].is_fixed = flag
        action.append("%s %s" % ("fix" if flag else "free", n))
    k = int(rs.choice([0, 1, 1, 2, 4, len(names)]))
    for n in rs.choice(names, size=min(k, len(names)), replace=False):
        prm = fast.params[n]
        if prm.range is None:               # (the other selection's object of that name has no range)
            continue
        lo, hi = prm.range[0].m_as(prm.units), prm.range[1].m_as(prm.units)
        v = rs.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo))
        for m in (fast, slow):
            m.params[n].value = v * m.params[n].units
        action.append(n)
    log.append(", ".join(action) or "nothing")
    try:
        a = template(fast)
        if rs.rand() < 0.15:
            a = template(fast)
        b = template(slow)
        ok = same(a, b)
        if last is not None and not np.array_equal(last, a[0]):
            moved += 1
        last = a[0]
        fresh_ok = True
        if step % 25 == 24 or not ok:
            fresh = build(False)
            if fast.param_selections:
                fresh.select_params(fast.param_selections)
            for prm in fast.params:
                q = fresh.params[prm.name]
                q.is_fixed = prm.is_fixed
                if prm.range is not None and not isinstance(prm.value, str):
                    q.value = prm.value
            c = template(fresh)
            fresh_ok = same(b, c)
            ok = ok and same(a, c)
            del fresh
        if not (ok and fresh_ok):
            bad += 1
            worst = float(np.max(np.abs(a[0] - b[0])) / max(np.abs(b[0]).max(), 1e-300))
            print("MISMATCH step %d (%s): plan vs protocol %.2e, protocol vs fresh maker %s | previous: %s"
                  % (step, log[-1], worst, "agree" if fresh_ok else "DIFFER", " / ".join(log[-4:-1])), flush=True)
    except Exception as e:  # pylint: disable=broad-except
        bad += 1
        print("ERROR step %d (%s): %s %s" % (step, log[-1], type(e).__name__, str(e)[:300]), flush=True)
    if step % 50 == 49:
        print("... %d steps, %d bad, %.0f s" % (step + 1, bad, time.time() - t0), flush=True)
print("fuzz_twins %s: %d steps, %d bad (the template moved in %d of them)" % (which, steps, bad, moved))
sys.exit(1 if bad else 0)
