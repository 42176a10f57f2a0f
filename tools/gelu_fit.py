"""Fit of the polynomial GELU of the large-batch GEMM epilogue (encoder.hip: gelu_poly2) and its error in fp32.
usage: python tools/gelu_fit.py   (CPU only)"""
import numpy as np
from scipy.special import erf

X0, TERMS = 4.0, 7


def fit(n=TERMS, x0=X0):
    xs = np.cos(np.pi * (np.arange(6000) + 0.5) / 6000) * x0 / 2 + x0 / 2
    a = np.stack([xs ** (2 * k + 1) for k in range(n)], 1)
    y = 0.5 * erf(xs / np.sqrt(2))
    base = np.maximum(xs, 0.05)                  # error metric |x| * |h error|: what GELU = x (0.5 + h) sees
    w = np.ones_like(xs)
    # h(x0) = 0.5 EXACTLY (the constraint is eliminated: c0 = (0.5 - sum_k c_k x0^(2k+1)) / x0): beyond the clamp GELU is
    # then x or 0 exactly, however large |x| is
    phi = np.stack([xs ** (2 * k + 1) - x0 ** (2 * k) * xs for k in range(1, n)], 1)
    tgt = y - 0.5 * xs / x0
    for _ in range(300):                         # iterated reweighting towards the minimax fit
        d, *_ = np.linalg.lstsq(phi * (w * base)[:, None], tgt * w * base, rcond=None)
        e = np.abs(phi @ d - tgt) * base
        w = w * (1 + 2 * e / e.max())
        w /= w.mean()
    c = np.concatenate([[(0.5 - sum(d[k - 1] * x0 ** (2 * k + 1) for k in range(1, n))) / x0], d])
    return c


def gelu_poly_f32(x, c, x0=X0):
    x = x.astype(np.float32)
    xc = np.clip(x, -x0, x0).astype(np.float32)
    t = (xc * xc).astype(np.float32)
    q = np.float32(c[-1]) * np.ones_like(t)
    for k in range(len(c) - 2, -1, -1):          # v_pk_fma_f32: one rounding per step (float64 holds the exact product)
        q = (q.astype(np.float64) * t.astype(np.float64) + np.float64(np.float32(c[k]))).astype(np.float32)
    h = (q * xc).astype(np.float32)
    half = (x * np.float32(0.5)).astype(np.float32)
    return (x.astype(np.float64) * h.astype(np.float64) + half.astype(np.float64)).astype(np.float32)


def snap(c, x0=X0):
    """Nudge the linear coefficient by a few fp32 ulps until the fp32 Horner evaluation gives q(x0^2) * x0 == 0.5 exactly."""
    c = np.array([np.float32(v) for v in c], np.float32)
    t = np.float32(x0 * x0)
    best = None
    for d in range(-200, 201):
        c0 = np.float32(c[0])
        for _ in range(abs(d)):
            c0 = np.nextafter(c0, np.float32(np.inf if d > 0 else -np.inf), dtype=np.float32)
        q = np.float32(c[-1])
        for k in range(len(c) - 2, 0, -1):
            q = np.float32(np.float64(q) * np.float64(t) + np.float64(c[k]))
        q = np.float32(np.float64(q) * np.float64(t) + np.float64(c0))
        if np.float32(q * np.float32(x0)) == np.float32(0.5) and (best is None or abs(d) < abs(best[0])):
            best = (d, c0)
    assert best is not None
    c[0] = best[1]
    return c.astype(np.float64)


if __name__ == "__main__":
    c = snap(fit())
    print("coefficients (x, x^3, ...):", [float(np.float32(v)) for v in c])
    x = np.concatenate([np.linspace(-8, 8, 400001), np.linspace(-1000, 1000, 20001)])
    ref = 0.5 * x * (1 + erf(x / np.sqrt(2)))
    print("h(x0) in fp32: %.9f" % float((gelu_poly_f32(np.array([X0]), c)[0] / np.float32(X0)) - np.float32(0.5)))
    print("max |GELU_poly - GELU_erf| over [-1000, 1000], fp32 evaluation: %.3e" % np.abs(gelu_poly_f32(x, c) - ref).max())
