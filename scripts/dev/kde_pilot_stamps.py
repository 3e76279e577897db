"""Read the per-wavefront stamps of kde_local_pilot_wave_kernel (development build, PISA_HIP_KDE_PILOT_STAMPS=<file>; the file holds
the LAST launch): when the wavefronts started, how long they waited for their inputs, how long the Horner schemes took.
    python scripts/dev/kde_pilot_stamps.py <file>"""
import sys, numpy as np
st = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4).astype(np.int64)
ran = st[:, 2] > 0
st = st[ran]
tick = 1e-2   # wall_clock64: 100 MHz -> us
t0 = st[:, 0].min()
start, mid, end = (st[:, 0] - t0) * tick, (st[:, 1] - t0) * tick, (st[:, 2] - t0) * tick
print("%d wavefronts ran; launch length %.1f us" % (len(st), end.max()))
print("start of a wavefront: median %.1f, p90 %.1f, last %.1f us" % (np.median(start), np.percentile(start, 90), start.max()))
print("inputs (block, coefficients -> LDS, coordinates): median %.2f, p90 %.2f, max %.2f us" % (np.median(mid - start), np.percentile(mid - start, 90), (mid - start).max()))
print("Horner schemes + store: median %.2f, p90 %.2f, max %.2f us" % (np.median(end - mid), np.percentile(end - mid, 90), (end - mid).max()))
h, e = np.histogram(start, bins=10, range=(0, max(start.max(), 1e-9)))
print("wavefront starts per tenth of the launch:", h.tolist())
live = np.array([((start <= t) & (end > t)).sum() for t in np.linspace(0, end.max(), 11)[:-1]])
print("wavefronts alive at the tenths:", live.tolist())
hw = st[:, 3]
xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
simd = (hwid >> 4) & 0x3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
dur = end - mid
ks, cnt = np.unique(key, return_counts=True)
print("compute units used: %d; wavefronts per unit: min %d, median %d, max %d" % (len(ks), cnt.min(), int(np.median(cnt)), cnt.max()))
mx = np.array([dur[key == k].max() for k in ks]); last = np.array([end[key == k].max() for k in ks])
order = np.argsort(-last)[:8]
print("units finishing last: (xcc, se, sh, cu) waves, longest Horner us, last end us")
for o in order:
    k = ks[o]; print("   ", (k // 256, (k // 32) % 8, (k // 16) % 2, k % 16), cnt[o], round(mx[o], 1), round(last[o], 1))
print("per XCC: waves, median Horner us, last end us:", [(int(x), int((xcc == x).sum()), round(float(np.median(dur[xcc == x])), 1), round(float(end[xcc == x].max()), 1)) for x in np.unique(xcc)])
ksimd = key * 4 + simd
ss, sc = np.unique(ksimd, return_counts=True)
print("wavefronts per SIMD: max %d, median %d; correlation (waves on the SIMD, Horner time): %.2f" % (sc.max(), int(np.median(sc)), np.corrcoef(sc[np.searchsorted(ss, ksimd)], dur)[0, 1]))
