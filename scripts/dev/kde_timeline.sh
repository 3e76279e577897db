# concurrency picture of one C3 evaluation (6 worker streams): union-busy time, idle gaps, per-kernel share
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ktl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktl -o k -- python3 scripts/dev/c3_probe.py 1e7 > gpurun_out/ktl.log 2>&1
grep '"it"' gpurun_out/ktl.log | cut -c1-40
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/ktl/k_kernel_trace.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last evaluation: from the 24th-last moments1 kernel onwards
starts = [e[0] for e in ev if "kde_moments1" in e[2]]
t_begin = starts[-24]
sel = [e for e in ev if e[0] >= t_begin - 300000]
t0, t1 = min(e[0] for e in sel), max(e[1] for e in sel)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("window %.2f ms, union busy %.2f ms, sum of kernel durations %.2f ms" % ((t1 - t0) / 1e6, busy / 1e6, sum(e - s for s, e, _ in sel) / 1e6))
acc = collections.Counter()
for s, e, n in sel:
    key = n.replace("void pisa::", "").replace("pisa::", "").split("(")[0][:40]
    if "rocprim" in n: key = "rocprim"
    if "at::native" in n: key = "torch"
    acc[key] += e - s
for k, v in acc.most_common(14): print("  %-42s %7.2f ms" % (k, v / 1e6))
# time with >= 1 heavy kernel (lattice / h2l / hermite / pilot / prep) running
heavy = [(s, e) for s, e, n in sel if any(x in n for x in ("kde_lattice_kernel", "kde_h2l", "kde_hermite_coef", "kde_local_pilot", "lattice_prep"))]
hb, cs, ce = 0, None, None
for s, e in sorted(heavy):
    if ce is None or s > ce:
        if ce is not None: hb += ce - cs
        cs, ce = s, e
    else: ce = max(ce, e)
hb += ce - cs
print("union of heavy kernels %.2f ms" % (hb / 1e6))
# the evaluation in slices of 250 us: heavy kernels running (mean concurrency), any kernel running (fraction)
SL = 250000
first_kde = min(s for s, e, n in sel if "kde_" in n)
first_heavy = min(s for s, e in heavy)
last_heavy = max(e for s, e in heavy)
print("evaluation starts %.2f ms before its first kde kernel; first heavy kernel at %.2f ms, last heavy kernel ends at %.2f ms, window ends %.2f ms"
      % ((first_kde - t0) / 1e6, (first_heavy - t0) / 1e6, (last_heavy - t0) / 1e6, (t1 - t0) / 1e6))
def cover(iv, a, b):
    return sum(max(0, min(e, b) - max(s, a)) for s, e in iv) / (b - a)
allk = [(s, e) for s, e, _ in sel]
lat = [(s, e) for s, e, n in sel if "kde_lattice_kernel" in n]
fgt = [(s, e) for s, e, n in sel if any(x in n for x in ("kde_h2l", "kde_hermite_coef", "kde_local_pilot"))]
print("slice   all  heavy  lattice  fgt   (mean number of kernels running)")
a = t0
while a < t1:
    b = min(a + SL, t1)
    print("%5.2f  %5.2f %5.2f %5.2f %5.2f" % ((a - t0) / 1e6, cover(allk, a, b), cover(heavy, a, b), cover(lat, a, b), cover(fgt, a, b)))
    a = b
PY
