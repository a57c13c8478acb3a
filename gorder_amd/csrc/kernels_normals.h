// kernels_normals.h — dynamic membrane normals (k_dyn_normals).
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

// ---- dynamic membrane normals ---------------------------------------------------------------------
// DynamicMembraneNormal::calculate_normal (normal.rs:160-199) for every molecule of every frame:
// cloud = "NormalHeads" atoms with 3-D (minimum-image) distance < radius from the molecule's head
// (pbc.rs:142-161, 321-350), normal = direction of least variance of the cloud (normal.rs:421-458).
// The cloud atoms go through the same cell list as the local-leaflet atoms (k_local_bin/scan/scatter,
// in-plane x-y cells whatever the membrane's orientation: the cells only prune); a wave per molecule
// accumulates count, sum d and sum d d^T of the minimum-image vectors d in f64 — the covariance does not
// depend on the origin — and lane 0 diagonalises it by cyclic Jacobi rotations in f64, the same operation
// sequence as the oracle.  nalgebra's f32 SVD cannot be restated bit for bit: this path is pinned by the
// reference's 4-decimal goldens only.  Sign convention: last non-zero component positive.
__device__ void sym3_smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22,
                                          double (&out)[3]) {
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double a[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        const double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-34 * dia)) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                const int r = 3 - p - q;
                const double apq = a[p][q], arp = a[r][p], arq = a[r][q];
                a[p][p] = a[p][p] - t * apq;
                a[q][q] = a[q][q] + t * apq;
                a[p][q] = a[q][p] = 0.0;
                a[r][p] = a[p][r] = c * arp - sn * arq;
                a[r][q] = a[q][r] = sn * arp + c * arq;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    const bool m1 = a[1][1] < a[0][0];
    const double d01 = m1 ? a[1][1] : a[0][0];
    const bool m2 = a[2][2] < d01;
    out[0] = m2 ? v[0][2] : (m1 ? v[0][1] : v[0][0]);
    out[1] = m2 ? v[1][2] : (m1 ? v[1][1] : v[1][0]);
    out[2] = m2 ? v[2][2] : (m1 ? v[2][1] : v[2][0]);
    const double lead = out[2] != 0.0 ? out[2] : (out[1] != 0.0 ? out[1] : out[0]);
    if (lead < 0.0) { out[0] = -out[0]; out[1] = -out[1]; out[2] = -out[2]; }
}

// block = 4 waves = 4 molecules; grid = (ceil(n_mol / 4), n_slab); a.heads = the molecules' normal heads,
// a.membrane = the cloud; out[(frame0 + s) * n_mol + m] = (nx, ny, nz, cloud size)
__global__ __launch_bounds__(256) void k_dyn_normals(LocalArgs a, float4 *__restrict__ out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t m = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t s = blockIdx.y;
    if (m >= a.n_mol_total) return;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    const float *hp = x + 3u * (size_t)a.heads[m];
    const float hx = hp[0], hy = hp[1], hz = hp[2];
    if (hx != hx) {
        if (lane == 0) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageSystem, 0, 0, m);
        return;
    }
    int bad = 0;
    uint32_t ha = 0, hb = 0;
    if (a.pbc) {
        const float wa = gm_wrap(hp[da], box[da], bad), wb = gm_wrap(hp[db], box[db], bad);
        ha = (uint32_t)fminf(fmaxf(floorf(wa / box[da] * (float)nca), 0.0f), (float)(nca - 1u));
        hb = (uint32_t)fminf(fmaxf(floorf(wb / box[db] * (float)ncb), 0.0f), (float)(ncb - 1u));
    }
    const uint32_t *cstart = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    const LocalRec *rec = reinterpret_cast<const LocalRec *>(a.trig) + (size_t)s * a.rec_stride;
    const float thr = a.radius_thr;
    const bool pbc = a.pbc != 0;
    const uint32_t n_rows = 2u * ka + 1u, n_cols = 2u * kb + 1u;
    const uint32_t a0 = (ha + nca - ka) % nca, b0 = (hb + ncb - kb) % ncb;
    const uint32_t b1 = min(b0 + n_cols, ncb), b2 = b0 + n_cols - b1;
    // records are (coordinate da, coordinate db, coordinate dim, -) = (x, y, z, -) for dim = 2
    double sx = 0.0, sy = 0.0, sz = 0.0, sxx = 0.0, sxy = 0.0, sxz = 0.0, syy = 0.0, syz = 0.0, szz = 0.0;
    uint32_t cnt = 0;
    for (uint32_t ia = 0; ia < n_rows; ia++) {
        const uint32_t row = ((a0 + ia) % nca) * ncb;
        for (uint32_t part = 0; part < 2u; part++) {
            const uint32_t q0 = part == 0 ? cstart[row + b0] : cstart[row];
            const uint32_t q1 = part == 0 ? cstart[row + b1] : cstart[row + b2];
            for (uint32_t q = q0 + lane; q < q1; q += 64u) {
                const LocalRec r = rec[q];
                float dx = r.x - hx, dy = r.y - hy, dz = r.z - hz;
                if (pbc) { dx = gm_min_image(dx, box[0], bad); dy = gm_min_image(dy, box[1], bad); dz = gm_min_image(dz, box[2], bad); }
                if ((dx * dx + dy * dy) + dz * dz < thr) {          // == sqrt(..) < radius (local_radius_threshold)
                    cnt += 1;
                    sx += (double)dx; sy += (double)dy; sz += (double)dz;
                    sxx += (double)dx * dx; sxy += (double)dx * dy; sxz += (double)dx * dz;
                    syy += (double)dy * dy; syz += (double)dy * dz; szz += (double)dz * dz;
                }
            }
        }
    }
    const double n = wave_total((double)cnt);
    sx = wave_total(sx); sy = wave_total(sy); sz = wave_total(sz);
    sxx = wave_total(sxx); sxy = wave_total(sxy); sxz = wave_total(sxz);
    syy = wave_total(syy); syz = wave_total(syz); szz = wave_total(szz);
    if (lane == 0) {
        float4 o = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), (float)n);
        if (n >= 3.0) {
            const double inv = 1.0 / n;
            double e[3];
            sym3_smallest_eigenvector(sxx - sx * sx * inv, sxy - sx * sy * inv, sxz - sx * sz * inv,
                                      syy - sy * sy * inv, syz - sy * sz * inv, szz - sz * sz * inv, e);
            const float fx = (float)e[0], fy = (float)e[1], fz = (float)e[2];
            const float len = __builtin_sqrtf((fx * fx + fy * fy) + fz * fz);     // Vector3D::to_unit
            o.x = fx / len; o.y = fy / len; o.z = fz / len;
        }
        out[(size_t)f * a.n_mol_total + m] = o;
    }
    if (bad) raise_box_range(a.err, f);
}

}  // namespace
