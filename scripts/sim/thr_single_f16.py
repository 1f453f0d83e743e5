"""Paper for k_threshold_mfma with ONE f16 per tap and ONE f16 for the column-filtered intermediate:
the rigorous error bound for a given choice of the column scale s and the classification scale S, and the number of
pixels the bound leaves undecided on the bench clip / a 4K frame / uniform noise.  numpy only (float64 model of the
kernel's arithmetic: f16 x f16 products are exact in float32, the accumulation is modelled as exact and its worst case
added to the bound).  Usage: python scripts/sim/thr_single_f16.py [--search]"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")


def gauss11():
    i = np.arange(11) - 5
    k = np.exp(-(i * i) / 8.0)
    k = (k / k.sum()).astype(np.float32)          # cv2: normalised in double, cast to float32
    return k.astype(np.float64)


def f16(x):
    return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)      # round to nearest even


def taps_error(w, scale):
    """sum_i |f16(scale w_i) / scale - w_i| over the eleven taps"""
    return np.abs(f16(w * scale) / scale - w).sum()


def bound(w, s, S, centre=128.0):
    """worst-case |kernel's (mean - b) - real (mean - b)| in gray levels, any image"""
    dc, dr = taps_error(w, s), taps_error(w, S / s)
    amp = max(centre, 255.0 - centre)
    v1max = s * amp * (1 + dc) * w.sum()
    half_ulp = 2.0 ** (np.floor(np.log2(v1max)) - 10) / 2
    rho = half_ulp / s
    acc = 32 * 2.0 ** -25 * 256 + 64 * 2.0 ** -25 * 2 * 256 * 1.0   # one float32 rounding per product, all lined up: ~2e-3
    cv2 = 22 * 2.0 ** -24 * 255                                       # cv2's own float32 chain against the real mean
    return amp * dc + rho + amp * dr / s * (1 + dc) + acc + cv2, dict(dc=dc, dr=dr, rho=rho, v1max=v1max)


def filt(img, taps, axis):
    pad = [(0, 0), (0, 0)]
    pad[axis] = (5, 5)
    p = np.pad(img, pad, mode="edge")
    out = np.zeros_like(img, dtype=np.float64)
    for i in range(11):
        sl = [slice(None), slice(None)]
        sl[axis] = slice(i, i + img.shape[axis])
        out += taps[i] * p[tuple(sl)]
    return out


def kernel_model(b, w, s, S, centre=128.0):
    """(mean - b) as the kernel would compute it, in gray levels"""
    c = b.astype(np.float64) - centre
    v1 = filt(c, f16(w * s), 0)                   # column pass, taps f16(s w)
    h = f16(v1)                                   # ONE f16, round to nearest
    x = filt(h, f16(w * S / s), 1)                # row pass, taps f16(S / s w)
    return x / S - c


def count(frames, w, s, S, eps, levels):
    from oracle import ysmr_oracle as oracle
    oracle.build()
    n_amb, worst = [], 0.0
    for g in frames:
        b = oracle.blur3(g)
        real = filt(filt(b.astype(np.float64), w, 0), w, 1) - b
        mine = kernel_model(b, w, s, S)
        worst = max(worst, float(np.abs(mine - real).max()))
        amb = np.zeros(b.shape, bool)
        for th in levels:
            amb |= np.abs(mine - th) < eps
        n_amb.append(int(amb.sum()))
    return n_amb, worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--search", action="store_true")
    ap.add_argument("--s", type=float, default=None)
    ap.add_argument("--S", type=float, default=None)
    args = ap.parse_args()
    w = gauss11()
    print("taps", w[:6], "sum", w.sum())
    if args.search:
        best = []
        for s in np.arange(0.90, 0.99951, 0.00005):
            dc = taps_error(w, s)
            if s * 128 * (1 + dc) >= 128:
                continue
            best.append((128 * dc + 2.0 ** -5 / s, s, dc))
        best.sort()
        print("column scale s: best by 128 dc + rho:", best[:5], "  s = 1 - 2^-10:", taps_error(w, 1 - 2.0 ** -10))
        s = best[0][1]
        cand = []
        for S in np.arange(16.0, 64.0, 1 / 32.0):    # S itself must be an f16 (the preset's tap): multiples of 1/32 below 64 are
            cand.append((taps_error(w, S / s), S))
        cand.sort()
        print("classification scale S: smallest row-tap error:", cand[:8])
    s = args.s or 0.9945          # what choose_scales (thr_mfma.hip) picks for these taps
    S = args.S or 39.375
    E, parts = bound(w, s, S)
    print(f"s = {s}, S = {S}: bound {E:.5f}  parts {parts}   EPS where the bf8 conversion decides = 1.875 / S = {1.875 / S:.5f}")
    from ysmr_amd.synth import SyntheticVideo
    rng = np.random.default_rng(0)
    levels = (-5.5, -7.5)
    for name, frames in (("bench clip, 3 frames", SyntheticVideo(922, 1228, 500, seed=0).frames(3)),
                         ("4K dense, 1 frame", SyntheticVideo(2160, 3840, 5000, seed=0).frames(1)),
                         ("uniform noise 400x1228", rng.integers(0, 256, (1, 400, 1228), dtype=np.uint8))):
        for eps in sorted({1.875 / S, 1 / 256}):
            n, worst = count(frames, w, s, S, eps, levels)
            print(f"{name}: EPS {eps:.5f}: undecided pixels per frame {n}; worst |model - real| {worst:.5f}")


if __name__ == "__main__":
    main()
