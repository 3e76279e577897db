"""Read the per-wavefront stamps of kde_lattice_kernel (development build, PISA_HIP_KDE_LATTICE_STAMPS=<file>):
lifetimes, steps and passes per wavefront and per sub-patch.  python scripts/dev/kde_stamps.py <file> [launch index]"""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
launches = []
i = raw.size   # from the end (the first launches of a process may come from several threads, interleaved)
while i > 0:
    ok = False
    for nw in (3072, 4096, 2048, 6144, 8192):
        j = i - 4 - 4 * nw
        if j >= 0 and int(raw[j]) == nw and int(raw[j + 3]) in (8, 16, 32, 64):
            npat, n, lg = int(raw[j + 1]), int(raw[j + 2]), int(raw[j + 3])
            launches.insert(0, (nw, npat, n, lg, raw[j + 4:i].reshape(nw, 4)))
            i = j
            ok = True
            break
    if not ok:
        break
nw, npat, n, lg, st = launches[which]
t0, t1, steps, pp = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64), st[:, 2].astype(np.int64), st[:, 3]
ran = t1 > 0
passes = (pp >> np.uint64(16)).astype(np.int64); patch = (pp & np.uint64(0xffff)).astype(np.int64)
tick = 1e-8   # wall_clock64: 100 MHz
T0 = t0[ran].min()
life = (t1 - t0)[ran] * tick * 1e6
print("launch %d of %d: %d wavefronts (%d ran), %d sub-patches, %d sources, LG %d" % (which, len(launches), nw, ran.sum(), npat, n, lg))
print("launch length %.1f us; wavefront lifetime mean %.1f, median %.1f, p90 %.1f, max %.1f us; start spread %.1f us"
      % ((t1[ran].max() - T0) * tick * 1e6, life.mean(), np.median(life), np.percentile(life, 90), life.max(), (t0[ran].max() - T0) * tick * 1e6))
print("steps per wavefront mean %.0f (min %d, max %d); passes mean %.0f (max %d); pass fraction %.2f"
      % (steps[ran].mean(), steps[ran].min(), steps[ran].max(), passes[ran].mean(), passes[ran].max(), passes[ran].sum() / max(steps[ran].sum(), 1)))
cost = steps * 28 + passes * 150
print("model cost (28 steps + 150 passes) per wavefront: mean %.0f, max %.0f; max / mean %.2f" % (cost[ran].mean(), cost[ran].max(), cost[ran].max() / cost[ran].mean()))
c = np.corrcoef(cost[ran], life)[0, 1]
print("correlation lifetime ~ model cost %.3f; us per 1000 model instructions: %.2f" % (c, 1e3 * life.sum() / cost[ran].sum()))
print("per sub-patch (first 20 by cost per wavefront): patch, waves, steps/wave, pass fraction, mean life us")
rows = []
for p in np.unique(patch[ran]):
    m = ran & (patch == p)
    rows.append((cost[m].mean(), p, m.sum(), steps[m].mean(), passes[m].sum() / max(steps[m].sum(), 1), ((t1 - t0)[m] * tick * 1e6).mean()))
for r in sorted(rows, reverse=True)[:12] + sorted(rows)[:6]:
    print("  patch %3d waves %4d steps %6.0f passfrac %.2f life %.1f cost %.0f" % (r[1], r[2], r[3], r[4], r[5], r[0]))
