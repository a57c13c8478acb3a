#!/usr/bin/env python3
"""How many heads does k_local_flags_rows decide WITHOUT their ring?  (kernels_leaflets.h, "a head its ring cannot change")

For sampled heads of a frame this restates the kernel's decision in numpy: the candidates are the atoms of the cells the
cylinder touches (approximated here by the atoms within r + one cell width of the head in the plane), the ring those that
are not within r - one cell width; T, A, B, the circular mean of the candidates and the three conditions (same image, head
near, |T - A/2| > sqrt(N B) / 2 + slack) as in the kernel.  It also checks that the interval really holds the members' sum.

  python tools/local_prune_probe.py            # synthetic CG membranes (flat, undulating) + the reference's cg fixture
Runs on the CPU; needs nothing but the repository (and tests/golden for the fixture)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gorder_amd import synthetic  # noqa: E402


def probe(xyz, box, heads, memb, r, name, sample=400):
    L = box[:2]
    Ln = box[2]
    cell = r / 7.0
    P, H = xyz[memb], xyz[heads]
    zmin, zmax = P[:, 2].min(), P[:, 2].max()
    rng = np.random.default_rng(0)
    n = decided = 0
    why = {"interval": 0, "image": 0, "near": 0}
    for i in rng.choice(len(heads), size=min(sample, len(heads)), replace=False):
        d = P[:, :2] - H[i, :2]
        d -= L * np.round(d / L)
        d2 = (d ** 2).sum(1)
        zh = H[i, 2]
        u = P[:, 2] - zh
        members = d2 < r * r
        cand = d2 < (r + cell) ** 2
        ring = cand & (d2 > (r - cell) ** 2)
        fn, fr, fi = cand.sum(), ring.sum(), (cand & ~ring).sum()
        T, A, B = u[cand].sum(), u[ring].sum(), (u[ring] ** 2).sum() * 1.02 + 1e-3 * ring.sum()
        mid, rad = T - 0.5 * A, 0.5005 * np.sqrt(fr * B)
        slack = 0.1 + fn * (1e-3 + 1e-6 * fn * Ln)
        true = u[members].sum()
        assert mid - rad - 1e-3 <= true <= mid + rad + 1e-3, "the interval does not hold the members' sum"
        ang = 2 * np.pi * P[cand, 2] / Ln
        cc, cs = np.cos(ang).sum(), np.sin(ang).sum()
        r_c = np.hypot(cc, cs)
        est = (np.arctan2(-cs, -cc) + np.pi) * Ln / (2 * np.pi)
        shift = zh - est
        shift -= Ln * np.round(shift / Ln)
        xr = (fr + 1.0) / max(r_c, 1e-30)
        emargin = 1e-4 * Ln + (xr + 0.5708 * xr ** 3) * 0.15916 * Ln
        same_image = fr + 1.0 < r_c and (zmin - zh) + shift > -Ln / 2 + emargin and (zmax - zh) + shift < Ln / 2 - emargin
        near = abs(mid) + rad < (Ln / 2 - 1e-3 * Ln) * fi
        interval = abs(mid) > rad + slack
        ok = fi > 0 and same_image and near and interval and (zmax - zmin) < 0.75 * Ln
        decided += ok
        why["interval"] += not interval
        why["image"] += not same_image
        why["near"] += not near
        if ok:
            assert (mid > 0) == (true > 0)
        n += 1
    print(f"{name}: {n} heads sampled, candidates ~{fn}, ring ~{fr}; decided without the ring {decided / n:.3f}"
          f"  (not decided for: interval {why['interval']}, image {why['image']}, near {why['near']}; frame z range"
          f" {zmax - zmin:.2f} of {Ln:.2f})")


def main():
    s = synthetic.cg_membrane(3072)
    xyz = np.asarray(s.frames(1, seed=1))[0].astype(np.float64)
    heads = np.concatenate([np.arange(s.n_atoms // 12) * 12 + 1])
    memb = np.arange(s.n_atoms)
    box = np.asarray(s.box, dtype=np.float64)
    probe(xyz, box, heads, memb, 2.5, "synthetic CG, 3072 lipids, flat (bench cg3k-local)")
    for amp in (0.5, 1.0, 2.0):
        x2 = xyz.copy()
        x2[:, 2] += amp * np.sin(2 * np.pi * x2[:, 0] / box[0]) * np.cos(2 * np.pi * x2[:, 1] / box[1])
        probe(x2, box, heads, memb, 2.5, f"  same, undulating by {amp} nm")
    try:
        from golden_util import METHODS, Fixture, cg_setup
        fx = Fixture("cg")
        tables, _, midx = cg_setup(fx, leaflets=METHODS["local"])
        for f in (0, len(fx.xyz) // 2, len(fx.xyz) - 1):
            X = fx.xyz[f][midx].astype(np.float64)
            bx = np.diag(np.asarray(fx.boxes[f]).reshape(3, 3)).astype(np.float64)
            hd = np.concatenate([np.asarray(m.heads) for m in tables.molecule_types])
            probe(X, bx, hd, np.asarray(tables.leaflets.membrane), float(tables.leaflets.radius),
                  f"the reference's cg.xtc, frame {f}")
    except Exception as e:      # noqa: BLE001
        print("reference fixture not available:", e)


if __name__ == "__main__":
    main()
