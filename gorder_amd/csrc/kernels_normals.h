// kernels_normals.h — dynamic membrane normals (k_dyn_cov, k_dyn_eigen).
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

// ---- dynamic membrane normals ---------------------------------------------------------------------
// DynamicMembraneNormal::calculate_normal (normal.rs:160-199) for every molecule of every frame:
// cloud = "NormalHeads" atoms with 3-D (minimum-image) distance < radius from the molecule's head
// (pbc.rs:142-161, 321-350), normal = direction of least variance of the cloud (normal.rs:421-458).
// The cloud atoms go through the same cell list as the local-leaflet atoms (k_local_bin/scan/scatter,
// in-plane x-y cells whatever the membrane's orientation: the cells only prune); k_dyn_cov accumulates count,
// sum d and sum d d^T of the minimum-image vectors d in f64 — the covariance does not depend on the origin — and
// k_dyn_eigen diagonalises it by cyclic Jacobi rotations in f64, the same operation sequence as the oracle.  nalgebra's f32 SVD cannot be restated bit for bit: this path is pinned by the
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

// count, sum d and sum d d^T of one molecule's cloud, as k_dyn_cov leaves them for k_dyn_eigen
struct DynCov { double n, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz; };

// Step 1 — the sums.  block = 256 threads = 4 waves = 16 molecules (a molecule = one DPP row of 16 lanes);
// grid = (ceil(n_mol / 16), n_slab); a.heads = the molecules' normal heads, a.membrane = the cloud.
// The candidates of a molecule are the records of the (2ka + 1) x (2kb + 1) cells around its head: lane i of the row takes
// row i of those cells — one contiguous run of records, two where the columns wrap around the box — and walks it, four
// loads in flight (a head has ~40 neighbours in ~80 cells: a wave per molecule striding over nine rows one after the
// other left 60 of 64 lanes idle and waited for eighteen dependent loads, 2.56 ms per 256 frames of 3072 lipids).
// Without periodic boundaries there is one cell: the sixteen lanes stride over its records.
__global__ __launch_bounds__(256) void k_dyn_cov(LocalArgs a, DynCov *__restrict__ cov) {
    const uint32_t lane = threadIdx.x & 63u, row = lane >> 4, sub = lane & 15u;
    const uint32_t m_raw = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + row;
    const uint32_t s = blockIdx.y;
    if ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u >= a.n_mol_total) return;       // the whole wave is past the last molecule
    const bool mol_ok = m_raw < a.n_mol_total;
    const uint32_t m = mol_ok ? m_raw : a.n_mol_total - 1u;                           // idle rows shadow the last molecule
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    const float *hp = x + 3u * (size_t)a.heads[m];
    const float hx = hp[0], hy = hp[1], hz = hp[2];
    const bool undefined = hx != hx;
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
    const uint32_t n_rows = 2u * ka + 1u, n_cols = 2u * kb + 1u;       // <= 9 (local_axis: reach <= kLocalFine)
    const uint32_t a0 = (ha + nca - ka) % nca, b0 = (hb + ncb - kb) % ncb;
    const uint32_t b1 = min(b0 + n_cols, ncb), b2 = b0 + n_cols - b1;
    // records are (coordinate da, coordinate db, coordinate dim) = (x, y, z) for dim = 2
    double sx = 0.0, sy = 0.0, sz = 0.0, sxx = 0.0, sxy = 0.0, sxz = 0.0, syy = 0.0, syz = 0.0, szz = 0.0;
    uint32_t cnt = 0;
    auto take = [&](const LocalRec r, bool valid) {
        float dx = r.x - hx, dy = r.y - hy, dz = r.z - hz;
        if (pbc) { dx = gm_min_image(dx, box[0], bad); dy = gm_min_image(dy, box[1], bad); dz = gm_min_image(dz, box[2], bad); }
        if (valid && (dx * dx + dy * dy) + dz * dz < thr) {          // == sqrt(..) < radius (local_radius_threshold)
            const double ex = (double)dx, ey = (double)dy, ez = (double)dz;       // (products and sums as f64 fmas: 9 + 3 instead of 6 + 6 + 9)
            cnt += 1;
            sx += ex; sy += ey; sz += ez;
            sxx = __builtin_fma(ex, ex, sxx); sxy = __builtin_fma(ex, ey, sxy); sxz = __builtin_fma(ex, ez, sxz);
            syy = __builtin_fma(ey, ey, syy); syz = __builtin_fma(ey, ez, syz); szz = __builtin_fma(ez, ez, szz);
        }
    };
    auto walk = [&](uint32_t q0, uint32_t q1, uint32_t first, uint32_t step) {      // records q0 + first, + step, ... below q1
        for (uint32_t q = q0 + first; q < q1; q += 4u * step) {
            LocalRec r[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) r[u] = rec[min(q + u * step, q1 - 1u)];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) take(r[u], q + u * step < q1);
        }
    };
    if (!undefined && mol_ok) {
        if (n_rows == 1u && n_cols == 1u) {         // one cell (no periodic boundaries): the row's lanes share its records
            walk(cstart[a0 * ncb + b0], cstart[a0 * ncb + b1], sub, 16u);
        } else if (sub < n_rows) {
            const uint32_t rw = ((a0 + sub) % nca) * ncb;
            walk(cstart[rw + b0], cstart[rw + b1], 0u, 1u);
            if (b2) walk(cstart[rw], cstart[rw + b2], 0u, 1u);
        }
    }
    // totals in lane 15 of the row (row shifts only)
    auto row_total = [](double v) {
        v = row_add_f64<0x111>(v); v = row_add_f64<0x112>(v); v = row_add_f64<0x114>(v);
        return row_add_f64<0x118>(v);
    };
    cnt = row_add_u32<0x111>(cnt); cnt = row_add_u32<0x112>(cnt); cnt = row_add_u32<0x114>(cnt); cnt = row_add_u32<0x118>(cnt);
    sx = row_total(sx); sy = row_total(sy); sz = row_total(sz);
    sxx = row_total(sxx); sxy = row_total(sxy); sxz = row_total(sxz);
    syy = row_total(syy); syz = row_total(syz); szz = row_total(szz);
    if (sub == 15u && mol_ok) {
        if (undefined) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageSystem, 0, 0, m);
        cov[(size_t)s * a.n_mol_total + m] = DynCov{undefined ? -1.0 : (double)cnt, sx, sy, sz, sxx, sxy, sxz, syy, syz, szz};
    }
    if (bad) raise_box_range(a.err, f);
}

// Step 2 — a thread per (slab frame, molecule): the covariance (it does not depend on the origin) diagonalised by the
// Jacobi sweeps above, 64 molecules per wave instead of one lane of every wave.
// out[(frame0 + s) * n_mol + m] = (nx, ny, nz, cloud size); an undefined head leaves its entry alone (the error is raised)
__global__ __launch_bounds__(256) void k_dyn_eigen(LocalArgs a, const DynCov *__restrict__ cov, float4 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)a.n_slab * a.n_mol_total) return;
    const uint32_t s = (uint32_t)(i / a.n_mol_total), m = (uint32_t)(i - (size_t)s * a.n_mol_total);
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    const DynCov c = cov[i];
    if (c.n < 0.0) return;
    float4 o = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), (float)c.n);
    if (c.n >= 3.0) {
        const double inv = 1.0 / c.n;
        double e[3];
        sym3_smallest_eigenvector(c.sxx - c.sx * c.sx * inv, c.sxy - c.sx * c.sy * inv, c.sxz - c.sx * c.sz * inv,
                                  c.syy - c.sy * c.sy * inv, c.syz - c.sy * c.sz * inv, c.szz - c.sz * c.sz * inv, e);
        const float fx = (float)e[0], fy = (float)e[1], fz = (float)e[2];
        const float len = __builtin_sqrtf((fx * fx + fy * fy) + fz * fz);     // Vector3D::to_unit
        o.x = fx / len; o.y = fy / len; o.z = fz / len;
    }
    out[(size_t)f * a.n_mol_total + m] = o;
}

}  // namespace
