"""dev tool (host only, numpy): replays fps_batch_kernel's rule -- take the runner-up wave's candidate as the NEXT sample when (a) the
winner's sample cannot reach the runner-up's wave, (b) its value exceeds the winner wave's second value, (c) it is unique -- on 16 Morton
waves of a cloud, checks the sequence against plain FPS and prints samples per exchange round for K = 2..4 candidates per round, with the
second value published eagerly or one round late.
    python tools/fps_batch_sim.py [uniform|lidar] [N] [npoint]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from pointrcnn_amd.rpn import synthetic_clouds, lidar_like_clouds

def spread(v):
    v = v & 0x3ff
    v = (v | (v << 16)) & 0x030000ff
    v = (v | (v << 8)) & 0x0300f00f
    v = (v | (v << 4)) & 0x030c30c3
    v = (v | (v << 2)) & 0x09249249
    return v

def morton_perm(p):
    lo = p.min(0); hi = p.max(0)
    sc = np.where(hi > lo, np.float32(1023.0) / (hi - lo), 0).astype(np.float32)
    q = ((p - lo) * sc).astype(np.uint32)
    m = (spread(q[:, 0]) << 2) | (spread(q[:, 2]) << 1) | spread(q[:, 1])
    key = (m.astype(np.uint64) << 32) | np.arange(len(p), dtype=np.uint64)
    return np.argsort(key, kind="stable")

def sim(p, npoint, NW=16, KMAX=4, lazy=True):
    N = len(p); perm = morton_perm(p); P = p[perm]          # Morton order
    per = N // NW
    wave_of = np.arange(N) // per
    lo = np.stack([P[w*per:(w+1)*per].min(0) for w in range(NW)]); hi = np.stack([P[w*per:(w+1)*per].max(0) for w in range(NW)])
    mind = np.full(N, 1e10, np.float32)
    val = np.full(NW, 1e10, np.float32); s = np.full(NW, 1e10, np.float32); cand = np.array([w*per for w in range(NW)])
    sknown = np.zeros(NW, bool)
    out = [int(np.where(perm == 0)[0][0])]   # first sample: original index 0 -> its morton position
    # apply first sample to all
    def apply(samples, waves):
        for w in waves:
            sl = slice(w*per, (w+1)*per)
            for smp in samples:
                d = ((P[sl] - P[smp])**2).astype(np.float32)
                d = (d[:, 0] + d[:, 1]) + d[:, 2]
                mind[sl] = np.minimum(mind[sl], d)
            m = mind[sl]
            # tie rule: lowest ORIGINAL index
            mx = m.max(); c = np.where(m == mx)[0]
            k = c[np.argmin(perm[sl][c])]
            val[w] = mx; cand[w] = w*per + k
            m2 = m.copy(); m2[k] = -1; s[w] = m2.max()
    apply([out[0]], range(NW))
    sknown[:] = not lazy
    rounds = 0; hist = np.zeros(KMAX+1, int); upd_hist = []
    def reach(smp):
        h = np.maximum(np.maximum(lo - P[smp], P[smp] - hi), 0).astype(np.float32)
        L = (h[:, 0]**2 + h[:, 1]**2) + h[:, 2]**2
        return L < val
    while len(out) < npoint:
        rounds += 1
        order = sorted(range(NW), key=lambda w: (-val[w], perm[cand[w]]))
        batch = [order[0]]; masks = reach(cand[order[0]]).copy()
        smax = s[order[0]] if sknown[order[0]] else np.float32(np.inf)
        for k in range(1, KMAX):
            if len(out) + len(batch) >= npoint: break
            w = order[k]
            nxt = val[order[k+1]] if k+1 < NW else -1
            if masks[w] or not (val[w] > smax) or not (val[w] > nxt): break
            batch.append(w); masks |= reach(cand[w])
            smax = max(smax, s[w] if sknown[w] else np.float32(np.inf))
        smp = [cand[w] for w in batch]
        out.extend(smp)
        hist[len(batch)] += 1
        ws = np.where(masks)[0]
        upd_hist.append(len(ws))
        apply(smp, ws)
        if lazy:
            sknown[:] = True; sknown[ws] = False
    return perm[np.array(out[:npoint])], rounds, hist, np.mean(upd_hist)

def plain(p, npoint):
    N = len(p); mind = np.full(N, 1e10, np.float32); out = [0]
    for j in range(1, npoint):
        d = ((p - p[out[-1]])**2).astype(np.float32); d = (d[:, 0] + d[:, 1]) + d[:, 2]
        mind = np.minimum(mind, d); out.append(int(np.argmax(mind)))
    return np.array(out)

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    npoint = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    cl = (synthetic_clouds if kind == "uniform" else lidar_like_clouds)(1, N)[0].numpy()
    ref = plain(cl, npoint)
    for lazy in (False, True):
        for K in (2, 3, 4):
            o, r, h, u = sim(cl, npoint, KMAX=K, lazy=lazy)
            print(kind, "lazy" if lazy else "eager", "K", K, "rounds", r, "samples/round %.2f" % ((npoint-1)/r), "hist", h.tolist(), "waves updating/round %.2f" % u, "exact", bool((o == ref).all()))
