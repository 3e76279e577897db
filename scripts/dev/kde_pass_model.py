"""CPU count of what kde_lattice_kernel executes for a patch shape (development; input: scripts/dev/kde_dump_estimator.py).

A wavefront owns a sub-patch of `sw` strips x `lpw` lines (sw * lpw = 64 / G lanes) and works G shares of 64 sources
side by side, one per lane group.  Counted per estimator: steps (one record per group), executed passes (any lane within
reach), lanes busy per pass, lane-instructions by the model  step_cost * steps + (S + 2 R) * 64 * passes.
    python scripts/dev/kde_pass_model.py gpurun_out/kde_dump/est0.npz [subsample]"""
import sys, numpy as np

def model(d, G, sw, R, tol=1e-14, sub=4, pairing=False, seg=0, S=60, T=14):
    ys = d["ys"].astype(np.float64).reshape(2, -1); s2 = d["s2"].astype(np.float64).reshape(-1)
    n = s2.size
    ya, yb = ys[0], ys[1]
    U, mean, origin, step, count = d["U"], d["mean"], d["origin"], d["step"], d["count"]
    rcut2 = 2.0 * np.log(1.0 / tol)
    dx0, dx1 = origin[0] - mean[0], origin[1] - mean[1]
    ya0 = U[0, 0] * dx0 + U[0, 1] * dx1; yb0 = U[1, 1] * dx1
    da = U[0, 0] * step[0]; sa = U[0, 1] * step[1]; db = U[1, 1] * step[1]
    n0, n1 = int(count[0]), int(count[1])
    C = R // 2
    lanes = 64 // G
    lpw = lanes // sw
    strips_a = -(-n0 // R)
    n_colblk = -(-strips_a // sw); n_rowblk = -(-n1 // lpw)
    ns = n // 64
    sh = slice(0, ns * 64)
    A = ya[sh].reshape(ns, 64); B = yb[sh].reshape(ns, 64); S2 = s2[sh].reshape(ns, 64)
    reach = np.sqrt(rcut2 / S2)
    box = np.stack([(A - reach).min(1), (A + reach).max(1), (B - reach).min(1), (B + reach).max(1)], 1)
    ext_lo, ext_hi = C * da, (R - 1 - C) * da
    tot = dict(steps=0, passes=0, busy=0, useful=0.0, second=0, segs=0)
    rng = np.random.RandomState(0)
    for rb in range(n_rowblk):
        for cb in range(n_colblk):
            t_f, j_f = cb * sw, rb * lpw
            t_l = min(t_f + sw, strips_a) - 1; j_l = min(j_f + lpw, n1) - 1
            pb = sorted([yb0 + j_f * db, yb0 + j_l * db])
            pa_lo = ya0 + min(j_f * sa, j_l * sa) + t_f * R * da
            pa_hi = ya0 + max(j_f * sa, j_l * sa) + ((t_l + 1) * R - 1) * da
            m = ~((box[:, 0] > pa_hi) | (box[:, 1] < pa_lo) | (box[:, 2] > pb[1]) | (box[:, 3] < pb[0]))
            lst = np.nonzero(m)[0]
            if sub > 1:   # subsample the list (every sub-th group of G shares), scale the counts
                k = (lst.size // G) * G
                lst = lst[:k].reshape(-1, G)[::sub].reshape(-1)
            if lst.size == 0:
                continue
            # lane geometry of the sub-patch
            ls = np.arange(lanes) % sw; ll = np.arange(lanes) // sw
            t = t_f + ls; j = j_f + ll
            live = (ll < lpw) & (t < strips_a) & (j < n1)
            ybl = yb0 + j * db
            yac = ya0 + j * sa + (t * R + C) * da
            a = A[lst][:, :, None]; b = B[lst][:, :, None]; s = S2[lst][:, :, None]
            xc = yac[None, None, :] - a
            dbb = ybl[None, None, :] - b
            dn = np.maximum(np.maximum(xc - ext_lo, -(xc + ext_hi)), 0.0)
            inr = live[None, None, :] & ((dbb * dbb + dn * dn) * s <= rcut2)      # [share, record, lane]
            if pairing:   # today's kernel: G = 1 lanes, two shares side by side (first half of the list with the second)
                h = (inr.shape[0] + 1) // 2
                Ain = inr[:h]; Bin = np.zeros_like(Ain); Bin[:inr.shape[0] - h] = inr[h:]
                first = Ain | Bin; both = Ain & Bin
                p1 = first.any(2); p2 = both.any(2)
                tot["steps"] += p1.size * sub
                tot["passes"] += (p1.sum() + p2.sum()) * sub
                tot["second"] += p2.sum() * sub
                tot["busy"] += (first.sum() + both.sum()) * sub
            else:
                k = (inr.shape[0] // G) * G
                g = inr[:k].reshape(-1, G, 64, lanes)          # [step group, G, record, lane]
                anyl = g.any(3).any(1)                         # [step group, record]
                tot["steps"] += anyl.size * sub
                tot["passes"] += anyl.sum() * sub
                tot["busy"] += g.sum() * sub
                if seg:   # wave-uniform trimming of the two chains in segments of `seg` points
                    # per lane: farthest needed point up (>= C) and down (< C) -> segments executed = max over lanes
                    kk = np.arange(R) - C
                    xs_ = xc[:k][..., None] + kk[None, None, None, :] * da                      # [share, record, lane, point]
                    ins = ((xs_ * xs_ + dbb[:k][..., None] ** 2) * s[:k][..., None] <= rcut2) & inr[:k][..., None]
                    ins = ins.reshape(-1, G, 64, lanes, R).transpose(0, 2, 1, 3, 4).reshape(-1, G * lanes, R)   # [step, lane of the wave, point]
                    anyp = ins.any(1)                                                            # [step, point]
                    up = anyp[:, C:]; dn = anyp[:, :C][:, ::-1]
                    far_up = np.where(up.any(1), R - C - np.argmax(up[:, ::-1], 1), 0)
                    far_dn = np.where(dn.any(1), C - np.argmax(dn[:, ::-1], 1), 0)
                    tot["segs"] += (np.ceil(far_up / seg).sum() + np.ceil(far_dn / seg).sum()) * seg * sub
            # useful pairs: lattice points of in-reach strips inside the disc (exact count on a subsample of records)
            rr = rng.randint(0, 64, size=min(8, 64))
            xs = xc[:, rr, :][..., None] + (np.arange(R) - C)[None, None, None, :] * da
            inside = (xs * xs + dbb[:, rr, :][..., None] ** 2) * s[:, rr, :][..., None] <= rcut2
            pts_ok = ((t * R)[:, None] + np.arange(R)[None, :] < n0)[None, None, :, :]
            tot["useful"] += float((inside & pts_ok & live[None, None, :, None]).sum()) * 64.0 / rr.size * sub
    cost = T * 64 * tot["steps"] + (S + 2 * R) * 64 * tot["passes"]
    if seg:
        cost = T * 64 * tot["steps"] + S * 64 * tot["passes"] + 2 * 64 * tot["segs"]
    return dict(G=G, sw=sw, lpw=lpw, R=R, tol=tol, patches=n_rowblk * n_colblk, steps_per_src=tot["steps"] / n, passes_per_src=tot["passes"] / n,
                lanes_busy=tot["busy"] / max(tot["passes"], 1), useful_pairs_per_src=tot["useful"] / n,
                lane_instr_per_src=cost / n, efficiency=2 * tot["useful"] / cost, points_per_pass=tot["segs"] / max(tot["passes"], 1))

if __name__ == "__main__":
    d = np.load(sys.argv[1])
    sub = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    print(model(d, 1, 1, 32, sub=sub, pairing=True))
    for G, sw, R in [(1, 1, 32), (2, 1, 32), (4, 1, 32), (8, 1, 32), (16, 1, 32), (4, 2, 32), (8, 2, 32), (4, 1, 16), (8, 1, 16), (8, 2, 16), (4, 2, 16), (4, 4, 16)]:
        print(model(d, G, sw, R, sub=sub))
    for tol in (1e-12, 1e-11):
        print(model(d, 8, 1, 32, tol=tol, sub=sub))
