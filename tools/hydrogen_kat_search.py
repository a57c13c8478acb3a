#!/usr/bin/env python3
"""Which operation order reproduces the reference's seven predicted hydrogen positions to ITS tolerance?

`tests_predict` (src/analysis/uaorder.rs:1113-1200) loads tests/files/ua.tpr and asserts 21 coordinates with
`assert_relative_eq!` at the default tolerance (f32::EPSILON absolute OR relative: about 1.2 ulp).  The thirteen atoms
involved have, in ua.tpr, exactly the f32 nearest to the three decimals ua_nobox.pdb prints (checked against the file's
big-endian f32 array, tools note in DESIGN §5), so the inputs are not in question.  The construction runs through
third-party code that is not in the checkout (groan_rs 0.11.2 `to_unit / shift / rotate`, nalgebra 0.34.0
`Unit::new_normalize / Rotation3::from_axis_angle`); the oracle's restatement comes within 2 ulp (4 ulp of the smallest
coordinate).  This script enumerates the plausible operation orders of those primitives and reports, per combination,
the number of coordinates outside the reference's tolerance and the largest error in ulps.  f32 arithmetic by numpy
scalars, sin / cos / acos by the host libm (as Rust's f32 methods on linux-gnu)."""
import ctypes
import ctypes.util
import itertools
import json
import sys

import numpy as np

f = np.float32
libm = ctypes.CDLL(ctypes.util.find_library("m"))
for name in ("sinf", "cosf", "acosf"):
    getattr(libm, name).argtypes = [ctypes.c_float]
    getattr(libm, name).restype = ctypes.c_float


def sinf(x): return f(libm.sinf(float(x)))
def cosf(x): return f(libm.cosf(float(x)))
def acosf(x): return f(libm.acosf(float(x)))


ATOMS = {11: (1.713, 2.717, 1.731), 12: (1.601, 2.675, 1.826), 13: (1.594, 2.754, 1.946),
         22: (1.193, 2.903, 2.586), 23: (1.118, 2.901, 2.720), 24: (1.075, 2.781, 2.774),
         31: (1.622, 2.525, 1.847), 38: (2.158, 2.258, 2.104), 39: (2.310, 2.254, 2.123),
         40: (2.346, 2.325, 2.254), 47: (3.052, 2.834, 2.149), 48: (3.172, 2.742, 2.176),
         49: (3.287, 2.820, 2.239)}
KATS = [("CH2", (38, 39, 40), [(2.3435528, 2.1503785, 2.1272178), (2.35857, 2.3045487, 2.039533)]),
        ("CH3", (48, 49, 47), [(3.3708375, 2.7527616, 2.257202), (3.254057, 2.8633823, 2.3334126), (3.3182635, 2.8995805, 2.1713943)]),
        ("CH1_UNSAT", (22, 23, 24), [(1.0985602, 2.994375, 2.7727659)]),
        ("CH1_SAT", (11, 31, 13, 12), [(1.5022101, 2.6938448, 1.7839708)])]
TET, TET_HALF, CH3_ANGLE, BOND, PI = f(1.910633), f(0.9553165), f(2.0943952), f(0.109), f(3.14159265358979323846)


def P(i): return tuple(f(c) for c in ATOMS[i])
def sub(a, b): return tuple(f(x - y) for x, y in zip(a, b))
def add(a, b): return tuple(f(x + y) for x, y in zip(a, b))
def neg(a): return tuple(f(-x) for x in a)
def cross(a, b): return (f(f(a[1] * b[2]) - f(a[2] * b[1])), f(f(a[2] * b[0]) - f(a[0] * b[2])), f(f(a[0] * b[1]) - f(a[1] * b[0])))


def length(a, mode):
    if mode == "sqrt((xx+yy)+zz)":
        return f(np.sqrt(f(f(f(a[0] * a[0]) + f(a[1] * a[1])) + f(a[2] * a[2]))))
    if mode == "sqrt(xx+(yy+zz))":
        return f(np.sqrt(f(f(a[0] * a[0]) + f(f(a[1] * a[1]) + f(a[2] * a[2])))))
    if mode == "f64 sum, f32 sqrt":
        return f(np.sqrt(f(float(a[0]) ** 2 + float(a[1]) ** 2 + float(a[2]) ** 2)))
    raise ValueError(mode)


def unit(a, mode, lmode):
    n = length(a, lmode)
    if mode == "v/len":
        return tuple(f(x / n) for x in a)
    if mode == "v*(1/len)":
        r = f(f(1.0) / n)
        return tuple(f(x * r) for x in a)
    raise ValueError(mode)


def rotation(u, angle):             # nalgebra Rotation3::from_axis_angle (unit axis)
    ux, uy, uz = u
    sqx, sqy, sqz = f(ux * ux), f(uy * uy), f(uz * uz)
    s, c = sinf(angle), cosf(angle)
    omc = f(f(1.0) - c)
    return ((f(sqx + f(f(f(1.0) - sqx) * c)), f(f(f(ux * uy) * omc) - f(uz * s)), f(f(f(ux * uz) * omc) + f(uy * s))),
            (f(f(f(ux * uy) * omc) + f(uz * s)), f(sqy + f(f(f(1.0) - sqy) * c)), f(f(f(uy * uz) * omc) - f(ux * s))),
            (f(f(f(ux * uz) * omc) - f(uy * s)), f(f(f(uy * uz) * omc) + f(ux * s)), f(sqz + f(f(f(1.0) - sqz) * c))))


def matvec(m, v, mode):
    if mode == "(m0 v0 + m1 v1) + m2 v2":
        return tuple(f(f(f(r[0] * v[0]) + f(r[1] * v[1])) + f(r[2] * v[2])) for r in m)
    if mode == "m0 v0 + (m1 v1 + m2 v2)":
        return tuple(f(f(r[0] * v[0]) + f(f(r[1] * v[1]) + f(r[2] * v[2]))) for r in m)
    raise ValueError(mode)


def shift(t, d, mode, umode, lmode):
    if mode == "t + unit(d)*b":
        u = unit(d, umode, lmode)
        return tuple(f(x + f(y * BOND)) for x, y in zip(t, u))
    if mode == "t + d*(b/len)":
        s = f(BOND / length(d, lmode))
        return tuple(f(x + f(y * s)) for x, y in zip(t, d))
    if mode == "t + (d*b)/len":
        n = length(d, lmode)
        return tuple(f(x + f(f(y * BOND) / n)) for x, y in zip(t, d))
    raise ValueError(mode)


def predict(kind, atoms, o):
    um, lm, mm, sm, num = o["to_unit"], o["len"], o["matvec"], o["shift"], o["new_normalize"]
    if kind == "CH3":
        h1, t, h2 = (P(i) for i in atoms)
        th1, th2 = sub(h1, t), sub(h2, t)
        ua = unit(cross(th2, th1), num, lm)
        hv1 = matvec(rotation(ua, TET), th1, mm)
        n1 = unit(th1, num, lm)
        return [shift(t, hv1, sm, um, lm), shift(t, matvec(rotation(n1, CH3_ANGLE), hv1, mm), sm, um, lm),
                shift(t, matvec(rotation(n1, f(-CH3_ANGLE)), hv1, mm), sm, um, lm)]
    if kind == "CH2":
        h1, t, h2 = (P(i) for i in atoms)
        th1, th2 = unit(sub(h1, t), um, lm), unit(sub(h2, t), um, lm)
        pn = cross(th2, th1)
        ra = unit(sub(th1, th2), um, lm)
        rv = cross(pn, ra)
        ura = unit(ra, num, lm)
        return [shift(t, matvec(rotation(ura, TET_HALF), rv, mm), sm, um, lm),
                shift(t, matvec(rotation(ura, f(-TET_HALF)), rv, mm), sm, um, lm)]
    if kind == "CH1_UNSAT":
        h1, t, h2 = (P(i) for i in atoms)
        th1, th2 = sub(h1, t), sub(h2, t)
        prod = f(f(f(th1[0] * th2[0]) + f(th1[1] * th2[1])) + f(th1[2] * th2[2]))
        c = f(prod / f(length(th1, lm) * length(th2, lm)))
        gamma = acosf(min(max(c, f(-1.0)), f(1.0)))
        ua = unit(cross(th1, th2), num, lm)
        return [shift(t, matvec(rotation(ua, f(PI - f(gamma / f(2.0)))), th2, mm), sm, um, lm)]
    h1, h2, h3, t = (P(i) for i in atoms)
    s = add(add(unit(sub(h1, t), um, lm), unit(sub(h2, t), um, lm)), unit(sub(h3, t), um, lm))
    return [shift(t, neg(s), sm, um, lm)]


OPTIONS = {"to_unit": ["v/len", "v*(1/len)"], "new_normalize": ["v/len", "v*(1/len)"],
           "len": ["sqrt((xx+yy)+zz)", "sqrt(xx+(yy+zz))", "f64 sum, f32 sqrt"],
           "matvec": ["(m0 v0 + m1 v1) + m2 v2", "m0 v0 + (m1 v1 + m2 v2)"],
           "shift": ["t + unit(d)*b", "t + d*(b/len)", "t + (d*b)/len"]}


def main():
    rows = []
    keys = list(OPTIONS)
    for combo in itertools.product(*(OPTIONS[k] for k in keys)):
        o = dict(zip(keys, combo))
        worst, outside, per = 0.0, 0, {}
        for kind, atoms, want in KATS:
            got = predict(kind, atoms, o)
            w = np.array(want, dtype=f)
            g = np.array(got, dtype=f)
            d = np.abs(g.astype(np.float64) - w.astype(np.float64))
            ok = (d <= np.finfo(f).eps) | (d <= np.maximum(np.abs(g), np.abs(w)).astype(np.float64) * np.finfo(f).eps)   # approx::relative_eq
            outside += int((~ok).sum())
            u = float((d / np.spacing(w).astype(np.float64)).max())
            per[kind] = u
            worst = max(worst, u)
        rows.append({**o, "coordinates_outside_the_reference_tolerance": outside, "max_ulp": worst, "max_ulp_per_kat": per})
    rows.sort(key=lambda r: (r["coordinates_outside_the_reference_tolerance"], r["max_ulp"]))
    for r in rows[:10] + rows[-3:]:
        print(r)
    oracle_row = next(r for r in rows if r["to_unit"] == "v/len" and r["new_normalize"] == "v/len" and r["len"] == "sqrt((xx+yy)+zz)"
                      and r["matvec"] == "(m0 v0 + m1 v1) + m2 v2" and r["shift"] == "t + unit(d)*b")
    print("the oracle's restatement:", oracle_row)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump({"n_coordinates": 21, "combinations": len(rows), "oracle": oracle_row, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
