#!/usr/bin/env python3
"""Derive the f32 polynomial coefficients used by the device trig kernels (gm_acosf / gm_cosf)
and by the oracle's MIRROR mode.  Remez-style refinement in float64 with mpmath-free numpy.

    asin(x) = x + x*z*R(z),  z = x*x in [0, 0.25]          (R: degree 4)
    sin(r)  = r + r*z*S(z),  z = r*r, |r| <= pi/4           (S: degree 2 -> up to r^7 ... use 3)
    cos(r)  = 1 - z/2 + z*z*C(z), |r| <= pi/4               (C: degree 2 -> up to r^8 ... use 3)

Prints C initialisers (hex floats) that are pasted into gorder_amd/csrc/gm_math.h and
oracle/gorder_oracle.c.  Deterministic; no randomness.
"""
import numpy as np

def remez(f, a, b, deg, weight=None, iters=30):
    """minimax polynomial of degree `deg` for f on [a,b] (absolute error, optional weight)."""
    n = deg + 2
    k = np.arange(n)
    x = 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * k / (n - 1))[::-1]
    xs = np.linspace(a, b, 20001)
    for _ in range(iters):
        w = np.ones_like(x) if weight is None else weight(x)
        A = np.zeros((n, n))
        for j in range(deg + 1):
            A[:, j] = x ** j
        A[:, deg + 1] = ((-1.0) ** k) / w
        sol = np.linalg.solve(A, f(x))
        c = sol[:deg + 1]
        err = (np.polyval(c[::-1], xs) - f(xs)) * (1.0 if weight is None else weight(xs))
        # locate extrema between sign changes
        sgn = np.sign(err)
        idx = np.flatnonzero(np.diff(sgn) != 0)
        bounds = np.concatenate(([0], idx + 1, [len(xs)]))
        newx = []
        for i in range(len(bounds) - 1):
            seg = slice(bounds[i], bounds[i + 1])
            j = np.argmax(np.abs(err[seg])) + bounds[i]
            newx.append(xs[j])
        if len(newx) != n:
            break
        newx = np.array(newx)
        if np.allclose(newx, x, rtol=0, atol=1e-12):
            break
        x = newx
    return c, np.max(np.abs(err))

def f_asin(z):
    z = np.maximum(z, 1e-300)
    s = np.sqrt(z)
    out = (np.arcsin(s) / s - 1.0) / z
    # series near 0 to avoid cancellation
    ser = 1.0 / 6 + z * (3.0 / 40 + z * (15.0 / 336 + z * (105.0 / 3456)))
    return np.where(z < 1e-4, ser, out)

def f_sin(z):
    z = np.maximum(z, 1e-300)
    r = np.sqrt(z)
    out = (np.sin(r) / r - 1.0) / z
    ser = -1.0 / 6 + z * (1.0 / 120 - z / 5040)
    return np.where(z < 1e-4, ser, out)

def f_cos(z):
    z = np.maximum(z, 1e-300)
    r = np.sqrt(z)
    out = (np.cos(r) - 1.0 + 0.5 * z) / (z * z)
    ser = 1.0 / 24 + z * (-1.0 / 720 + z / 40320)
    return np.where(z < 1e-3, ser, out)

def show(name, c):
    print("/* %s */" % name)
    for i, v in enumerate(c):
        f = np.float32(v)
        print("    %s  /* %.10e */," % (float(f).hex() + "f", float(f)))

if __name__ == "__main__":
    c, e = remez(f_asin, 0.0, 0.25, 5)
    print("asin R deg5 max err %.3e (x z x<=0.125 scale)" % e); show("ASIN_R", c)
    c, e = remez(f_asin, 0.0, 0.25, 4)
    print("asin R deg4 max err %.3e" % e); show("ASIN_R4", c)
    q = (np.pi / 4) ** 2
    c, e = remez(f_sin, 0.0, q, 2)
    print("sin S deg2 max err %.3e" % e); show("SIN_S", c)
    c, e = remez(f_sin, 0.0, q, 3)
    print("sin S deg3 max err %.3e" % e); show("SIN_S3", c)
    c, e = remez(f_cos, 0.0, q, 2)
    print("cos C deg2 max err %.3e" % e); show("COS_C", c)
    c, e = remez(f_cos, 0.0, q, 3)
    print("cos C deg3 max err %.3e" % e); show("COS_C3", c)
    for name, v in (("pio2", np.pi / 2), ("pi", np.pi)):
        hi = np.float32(v)
        lo = np.float32(v - float(hi))
        print(name, float(hi).hex(), float(lo).hex(), float(hi), float(lo))
