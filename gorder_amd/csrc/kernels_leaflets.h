// kernels_leaflets.h — leaflet classifiers (global, individual, local) and the per-frame shapes of the geometry selection.
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

// ---- leaflets ------------------------------------------------------------------------------
struct LeafletArgs {
    const float *xyz;
    const float *box9;
    uint32_t n_atoms;
    const uint32_t *aframes;   // [n_assign] local frame index of each assignment frame
    uint32_t row0;             // first output row
    uint8_t *aflags;           // [rows][n_mol_total]
    float *adist;              // [n_mol_total] signed distance of the LAST assignment frame (debug/tests)
    uint32_t n_mol_total;
    const uint32_t *heads;     // [n_mol_total] head atom per molecule
    const uint32_t *membrane;  // Global: membrane atom list
    uint32_t n_membrane;
    const uint32_t *methyl_begin;  // Individual: [n_mol_total+1] ranges into methyl_atoms
    const uint32_t *methyl_atoms;
    uint32_t dim;
    int flip, pbc;
    uint32_t *err;
    const uint8_t *skip;       // Global, after a speculative batch: [n_frames] 1 = this frame's centre is known already, leave
    uint32_t n_assign;         // Global: assignment frames of the launch (the grid may be smaller: workgroups take them in turn)
};

// cos / sin of 2*pi*u by the hardware v_cos_f32 / v_sin_f32 (argument in revolutions, ~1e-6 absolute
// error).  Used only for the Bai-Breen circular-mean ESTIMATE: the estimate merely anchors the
// minimum-image refinement pass that produces the centre, so its last digits do not matter.
__device__ __forceinline__ void fast_sincos_rev(float u, float *sn, float *cs);
// sin / cos of the angle 2 pi wrap(z) / L of a record's normal coordinate (they only make the ESTIMATE of a circular
// mean, which anchors an image choice: a multiplication by 1 / L is enough, and a coordinate outside the box gives an
// angle off by whole turns — the same sine and cosine)
__device__ __forceinline__ void local_trig(float z, float inv_L, float *sn, float *cs) {
    fast_sincos_rev(z * inv_L, sn, cs);          // (v_sin / v_cos take revolutions and reduce the range themselves)
}
__device__ __forceinline__ void fast_sincos_rev(float u, float *sn, float *cs) {
    *sn = __builtin_amdgcn_sinf(u);
    *cs = __builtin_amdgcn_cosf(u);
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// deterministic block reduction: butterfly inside each wave, then every thread adds the <= 16 wave
// totals in the same order (2 barriers)
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    const uint32_t wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
    if ((threadIdx.x & 63u) == 0) scratch[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (uint32_t w = 0; w < n_waves; w++) r += scratch[w];
    __syncthreads();
    return r;
}

// two sums for the price (barriers) of one; scratch holds 2 x 16 doubles
__device__ __forceinline__ void block_sum2(double &v0, double &v1, double *scratch) {
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    const uint32_t wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
    if ((threadIdx.x & 63u) == 0) { scratch[wave] = v0; scratch[16 + wave] = v1; }
    __syncthreads();
    double r0 = 0.0, r1 = 0.0;
    for (uint32_t w = 0; w < n_waves; w++) { r0 += scratch[w]; r1 += scratch[16 + w]; }
    __syncthreads();
    v0 = r0; v1 = r1;
}

// ---- per-frame shapes of the geometry selection: GeometrySelection::init_reference (geometry.rs:192-210)
// + construct_shape (geometry.rs:328-357, 422-451, 507-514).  One block per frame; a group reference needs
// the centre of geometry of the group (refined Bai-Breen, like the global membrane centre).
struct GeomArgs {
    const float *xyz;
    const float *box9;
    uint32_t n_atoms;
    int pbc;
    uint32_t kind, reference, orientation;
    float point[3];
    const uint32_t *group;
    uint32_t n_group;
    float xdim[2], ydim[2], zdim[2], radius, span[2], structure_box[3];
    float *shapes;   // [n_frames][8]
    uint32_t *err;
};

__global__ __launch_bounds__(256) void k_geom_shapes(GeomArgs g) {
    __shared__ double scratch[256];
    const uint32_t f = blockIdx.x;
    float box[3] = {1.0f, 1.0f, 1.0f};
    if (g.pbc) { const float *b = g.box9 + 9 * (size_t)f; box[0] = b[0]; box[1] = b[4]; box[2] = b[8]; }
    int bad = 0;
    float ref[3] = {g.point[0], g.point[1], g.point[2]};
    float shape_box[3] = {box[0], box[1], box[2]};
    if (g.reference == GORDER_GEOMREF_BOX_CENTER) {
        for (int d = 0; d < 3; d++) ref[d] = box[d] / 2.0f;
    } else if (g.reference == GORDER_GEOMREF_GROUP) {
        const float *x = g.xyz + (size_t)f * g.n_atoms * 3u;
        float est[3] = {0.0f, 0.0f, 0.0f};
        if (g.pbc) {
            double sc[3] = {0, 0, 0}, ss[3] = {0, 0, 0};
            for (uint32_t i = threadIdx.x; i < g.n_group; i += blockDim.x) {
                const float *p = x + 3u * (size_t)g.group[i];
                for (int d = 0; d < 3; d++) {
                    float sn, cs;
                    fast_sincos_rev(gm_wrap(p[d], box[d], bad) / box[d], &sn, &cs);
                    sc[d] += (double)cs;
                    ss[d] += (double)sn;
                }
            }
            for (int d = 0; d < 3; d++) {
                const double tc = block_sum(sc[d], scratch), ts = block_sum(ss[d], scratch);
                est[d] = (atan2f(-(float)ts, -(float)tc) + 3.1415927f) / (6.2831855f / box[d]);
            }
        }
        // Refinement = plain centre of the atoms' images nearest to the estimate (the atom's own coordinate
        // shifted by whole box lengths: the estimate only picks the image), summed in f32 in atom order like
        // the reference does: a sample 1 ulp from the shape's surface depends on the last bit
        // of this centre (the golden aa_order_sphere_dynamic.yaml has one), so the order of the sum is
        // part of the result.  One thread per frame does it; reference groups are small (a residue, a
        // protein), and the estimate above only selects the images, its own last bits do not matter.
        if (threadIdx.x == 0) {
            float acc[3] = {0.0f, 0.0f, 0.0f};
            for (uint32_t i = 0; i < g.n_group; i++) {
                const float *p = x + 3u * (size_t)g.group[i];
                for (int d = 0; d < 3; d++) acc[d] += g.pbc ? gm_nearest_image(p[d], est[d], box[d], bad) : p[d];
            }
            for (int d = 0; d < 3; d++) {
                const float c = acc[d] / (float)g.n_group;
                ref[d] = g.pbc ? gm_wrap(c, box[d], bad) : c;
            }
        }
    } else {
        for (int d = 0; d < 3; d++) shape_box[d] = g.structure_box[d];   // fixed point: built once, structure box
    }
    if (threadIdx.x == 0) {
        const float anchor = g.pbc ? 0.0f : -3.40282347e+38f;   // get_infinite_span, pbc.rs:236-240, 392-396
        const float inf = __builtin_inff();
        float sh[8] = {ref[0], ref[1], ref[2], 0.0f, 0.0f, 0.0f, g.radius, 0.0f};
        if (g.kind == GORDER_GEOM_CUBOID) {
            const float *dims[3] = {g.xdim, g.ydim, g.zdim};
            for (int d = 0; d < 3; d++) {
                if (dims[d][0] == -inf && dims[d][1] == inf) { sh[d] = anchor; sh[3 + d] = inf; }
                else { sh[d] = ref[d] + dims[d][0]; sh[3 + d] = dims[d][1] - dims[d][0]; }
            }
        } else if (g.kind == GORDER_GEOM_CYLINDER) {
            const int o = (int)g.orientation;
            if (g.span[0] == -inf && g.span[1] == inf) { sh[o] = anchor; sh[7] = inf; }
            else { sh[o] = ref[o] + g.span[0]; sh[7] = g.span[1] - g.span[0]; }
        }
        if (g.pbc) for (int d = 0; d < 3; d++) sh[d] = gm_wrap(sh[d], shape_box[d], bad);
        for (int k = 0; k < 8; k++) g.shapes[8 * (size_t)f + k] = sh[k];
    }
    if (bad) raise_box_range(g.err, f);
}

// One block per assignment frame: refined Bai-Breen centre of the membrane group
// (leaflets.rs:186-197 -> groan_rs group_get_center) followed by common_identify_leaflet
// (leaflets.rs:711-732) for every molecule.  Per-thread f32 partial sums are combined in f64 (the
// reference sums f32 sequentially; only the sign of head - centre is consumed).
__device__ __forceinline__ void leaflets_global_frame(const LeafletArgs &a, uint32_t bi) {
    __shared__ double scratch[16];
    __shared__ float s_center;
    const uint32_t f = a.aframes[bi];
    if (a.skip && a.skip[f]) return;                                  // (uniform over the workgroup)
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    const uint32_t dn = a.dim;
    float L = 1.0f;
    if (a.pbc) L = a.box9[9 * (size_t)f + 4 * dn];
    int bad = 0;
    // Only the component of the centre along the normal is consumed (leaflets.rs:725); the other two
    // matter only through the reference's NaN check (leaflets.rs:190-192): a non-finite coordinate of
    // any membrane atom makes the centre NaN -> InvalidGlobalMembraneCenter.
    float nonfinite = 0.0f;   // stays 0 while every coordinate is finite (x - x is 0 or NaN)
    float est = 0.0f;
    // the first KEEP normal-coordinates of each thread stay in registers for the second pass
    constexpr int KEEP = 32;
    float keep[KEEP];
    const uint32_t nthr = blockDim.x;
    float sc = 0.0f, ss = 0.0f;   // per-thread partials (<= n/1024 terms), combined in f64 below
    const float inv = a.pbc ? 1.0f / L : 0.0f;
    // batches of 8 atoms: the 24 loads of a batch are issued back to back (index clamped: lanes past the
    // end re-read the last atom and are masked out), then the batch is consumed
#pragma unroll
    for (int kb = 0; kb < KEEP; kb += 8) {
        float nf8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t i = threadIdx.x + (uint32_t)(kb + k) * nthr;
            const float *p = x + 3u * (size_t)a.membrane[i < a.n_membrane ? i : a.n_membrane - 1u];
            const float px = p[0], py = p[1], pz = p[2];
            nf8[k] = ((px - px) + (py - py)) + (pz - pz);
            keep[kb + k] = dn == 0 ? px : (dn == 1 ? py : pz);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool valid = threadIdx.x + (uint32_t)(kb + k) * nthr < a.n_membrane;
            nonfinite += valid ? nf8[k] : 0.0f;
            if (a.pbc) {
                float sn, cs;
                fast_sincos_rev(gm_wrap(keep[kb + k], L, bad) * inv, &sn, &cs);
                sc += valid ? cs : 0.0f;
                ss += valid ? sn : 0.0f;
            }
        }
    }
    for (uint32_t i = threadIdx.x + (uint32_t)KEEP * nthr; i < a.n_membrane; i += nthr) {   // very large groups
        const float *p = x + 3u * (size_t)a.membrane[i];
        const float px = p[0], py = p[1], pz = p[2];
        nonfinite += ((px - px) + (py - py)) + (pz - pz);
        if (a.pbc) {
            float sn, cs;
            fast_sincos_rev(gm_wrap(dn == 0 ? px : (dn == 1 ? py : pz), L, bad) * inv, &sn, &cs);
            sc += cs;
            ss += sn;
        }
    }
    if (a.pbc) {
        const double tc = block_sum((double)sc, scratch), ts = block_sum((double)ss, scratch);
        est = (atan2f(-(float)ts, -(float)tc) + 3.1415927f) / (6.2831855f / L);
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const uint32_t i = threadIdx.x + (uint32_t)k * nthr;
        if (i < a.n_membrane) {
            const float dx = keep[k] - est;
            acc += a.pbc ? gm_min_image(dx, L, bad) : dx;
        }
    }
    for (uint32_t i = threadIdx.x + (uint32_t)KEEP * nthr; i < a.n_membrane; i += nthr) {
        const float dx = x[3u * (size_t)a.membrane[i] + dn] - est;
        acc += a.pbc ? gm_min_image(dx, L, bad) : dx;
    }
    const double tot = block_sum((double)acc, scratch);
    const double nf = block_sum((double)nonfinite, scratch);
    if (threadIdx.x == 0) {
        float c = est + (float)(tot / (double)a.n_membrane);
        if (a.pbc) c = gm_wrap(c, L, bad);
        if (c != c || nf != 0.0 || a.n_membrane == 0) {
            raise_error(a.err, GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER, f, kStageSystem);
            c = __builtin_nanf("");
        }
        s_center = c;
    }
    __syncthreads();
    const float cdim = s_center;
    uint8_t *row = a.aflags + (size_t)(a.row0 + bi) * a.n_mol_total;
    const bool last = bi + 1 == a.n_assign;
    for (uint32_t m = threadIdx.x; m < a.n_mol_total; m += blockDim.x) {
        const float hp = x[3u * (size_t)a.heads[m] + dn];
        float d = hp - cdim;
        if (a.pbc) d = gm_min_image(d, L, bad);
        row[m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
        if (last && a.adist) a.adist[m] = d;
    }
    if (bad) raise_box_range(a.err, f);
}
// grid = the assignment frames, or fewer workgroups that take them in turn (the launch behind a speculative batch, where
// nearly every frame is skipped: a few hundred workgroups look at the flags instead of thousands being dispatched to leave)
// (TURNS only for that launch: with the loop around it the frame's arithmetic spills — 67 registers —, and the ordinary
// launch, a workgroup per frame, need not pay for that)
template <bool TURNS>
__global__ __launch_bounds__(1024) void k_leaflets_global(LeafletArgs a) {
    if (!TURNS) {
        leaflets_global_frame(a, blockIdx.x);
        return;
    }
    for (uint32_t bi = blockIdx.x; bi < a.n_assign; bi += gridDim.x) {
        leaflets_global_frame(a, bi);
        __syncthreads();
    }
}

// The same classifier for the usual case that the membrane group is EVERY atom of the frame, in order (all
// lipid atoms are decoded and "@membrane" selects them all): the frame is then one contiguous run of 3N floats,
// read ONCE with coalesced 16-byte loads instead of three strided dwords per atom through an index list.
// Float g of the frame is component g mod 3 of atom g / 3; a 16-byte chunk holds one or two normal components.
//
// One pass: next to the circular sums (estimate) every thread also sums u = minimum image of (z - z_ref), z_ref
// = the normal coordinate of the first molecule's head, and tracks min / max of u.  Once the estimate is known:
// if every atom's image relative to z_ref is also its image relative to the estimate (u_min, u_max inside the
// half box around it — any membrane thinner than half the box), then sum MI(z - est) = sum u + n (z_ref - est)
// and the refinement needs no second look at the data; otherwise the frame is read again.  256-thread blocks,
// nothing kept per atom: several frames per CU overlap their load and reduction phases.
__device__ __forceinline__ void block_minmax(float &lo, float &hi, float *scratch /* 2 x 16 */) {
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    const uint32_t wave = threadIdx.x >> 6, n_waves = (blockDim.x + 63u) >> 6;
    if ((threadIdx.x & 63u) == 0) { scratch[wave] = lo; scratch[16 + wave] = hi; }
    __syncthreads();
    for (uint32_t w = 0; w < n_waves; w++) { lo = fminf(lo, scratch[w]); hi = fmaxf(hi, scratch[16 + w]); }
    __syncthreads();
}

__device__ __forceinline__ void leaflets_global_contig_frame(const LeafletArgs &a, uint32_t bi) {
    __shared__ double scratch[32];
    __shared__ float fscratch[32];
    __shared__ float s_center;
    const uint32_t f = a.aframes[bi];
    if (a.skip && a.skip[f]) return;                                  // (uniform over the workgroup)
    const uint32_t dn = a.dim;
    float L = 1.0f;
    if (a.pbc) L = a.box9[9 * (size_t)f + 4 * dn];
    int bad = 0;
    const size_t start = (size_t)f * a.n_atoms * 3u;                 // first float of the frame
    const uint32_t sh = (uint32_t)(start & 3u), n_float = 3u * a.n_atoms;
    const uint32_t n4 = (sh + n_float + 3u) >> 2;
    const v4f *src = reinterpret_cast<const v4f *>(a.xyz + (start - sh));   // xyz is 16-byte aligned
    const float *x = a.xyz + start;
    const bool pbc = a.pbc != 0;
    const float inv = pbc ? 1.0f / L : 0.0f;
    const float zref = a.n_mol_total ? x[3u * (size_t)a.heads[0] + dn] : 0.0f;
    // chunk c = tid + 256 k starts at frame float 4c - sh; its element j is a normal component iff
    // (4c - sh + j) % 3 == dn, i.e. j % 3 == (dn + sh - tid - k) % 3 (4 and 256 are 1 mod 3)
    uint32_t r = (dn + sh + 3u * 256u - threadIdx.x) % 3u;            // j0 of this thread's chunk k; k -> k+1: r -> r-1
    float nonfinite = 0.0f, sc = 0.0f, ss = 0.0f, su = 0.0f, ulo = 3.0e38f, uhi = -3.0e38f;
    auto take = [&](float z, bool ok) {
        z = ok ? z : zref;               // a masked lane holds an in-plane coordinate: keep it out of the image search
        if (pbc) {
            bad |= (ok && !(z <= 9.0f * L && z >= -8.0f * L)) ? 1 : 0;   // gm_wrap would give up: coordinate far outside
            float sn, cs;
            fast_sincos_rev(z * inv, &sn, &cs);                             // v_sin / v_cos reduce the range themselves
            sc += ok ? cs : 0.0f;
            ss += ok ? sn : 0.0f;
        }
        const float u = pbc ? gm_min_image(z - zref, L, bad) : z - zref;
        su += ok ? u : 0.0f;
        ulo = ok ? fminf(ulo, u) : ulo;
        uhi = ok ? fmaxf(uhi, u) : uhi;
    };
    auto chunk = [&](const v4f v, uint32_t c, uint32_t j0) {
        float za = v.x;
        za = j0 == 1 ? v.y : za;
        za = j0 == 2 ? v.z : za;
        const uint32_t e0 = 4u * c;                                   // position of v.x counted from src
        if (e0 >= sh && e0 + 4u <= sh + n_float) {                    // every float of the chunk belongs to the frame
            const float t = (v.x + v.y) + (v.z + v.w);
            nonfinite += t - t;                                       // NaN / inf in any of them survives the sum
            take(za, true);
            take(v.w, j0 == 0);
        } else {                                                       // first / last chunk of the frame
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const bool in = e0 + j >= sh && e0 + j < sh + n_float;
                nonfinite += in ? e[j] - e[j] : 0.0f;
            }
            take(za, e0 + j0 >= sh && e0 + j0 < sh + n_float);
            take(v.w, j0 == 0 && e0 + 3u >= sh && e0 + 3u < sh + n_float);
        }
    };
    // Three consecutive chunks of a thread (c, c + 256, c + 512) hold exactly four normal components: one each at
    // j0 = r, r - 1, r - 2 (mod 3) and the .w of the chunk whose j0 is 0, i.e. chunk number r — and r is the same
    // again after three chunks.  Taking them as a group does four samples' worth of arithmetic instead of six
    // half-masked ones.  Two groups per iteration: six loads in flight.
    const uint32_t j1 = r == 0 ? 2u : r - 1u, j2 = j1 == 0 ? 2u : j1 - 1u;
    auto pick = [](const v4f v, uint32_t j0) {
        float z = v.x;
        z = j0 == 1 ? v.y : z;
        z = j0 == 2 ? v.z : z;
        return z;
    };
    auto group = [&](const v4f v0, const v4f v1, const v4f v2, uint32_t c0) {
        if (4u * c0 >= sh && 4u * (c0 + 512u) + 4u <= sh + n_float) {      // all twelve floats belong to the frame
            const float t = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)) + ((v2.x + v2.y) + (v2.z + v2.w));
            nonfinite += t - t;
            take(pick(v0, r), true);
            take(pick(v1, j1), true);
            take(pick(v2, j2), true);
            take(r == 0 ? v0.w : (r == 1 ? v1.w : v2.w), true);
        } else {
            chunk(v0, c0, r);
            chunk(v1, c0 + 256u, j1);
            chunk(v2, c0 + 512u, j2);
        }
    };
    uint32_t c = threadIdx.x;
    for (; c + 5u * 256u < n4; c += 6u * 256u) {
        v4f v[6];
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) v[k] = __builtin_nontemporal_load(src + c + k * 256u);
        group(v[0], v[1], v[2], c);
        group(v[3], v[4], v[5], c + 3u * 256u);
    }
    for (; c + 2u * 256u < n4; c += 3u * 256u) {
        v4f v[3];
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) v[k] = __builtin_nontemporal_load(src + c + k * 256u);
        group(v[0], v[1], v[2], c);
    }
    for (; c < n4; c += 256u) {
        chunk(__builtin_nontemporal_load(src + c), c, r);
        r = r == 0 ? 2u : r - 1u;
    }
    double tc = (double)sc, ts = (double)ss;
    block_sum2(tc, ts, scratch);
    double tu = (double)su, nf = (double)nonfinite;
    block_sum2(tu, nf, scratch);
    block_minmax(ulo, uhi, fscratch);
    float center;
    if (!pbc) {
        center = zref + (float)(tu / (double)a.n_atoms);
    } else {
        const float est = (atan2f(-(float)ts, -(float)tc) + 3.1415927f) / (6.2831855f / L);
        const float shift = gm_min_image(zref - est, L, bad);         // z_ref as seen from the estimate
        const float half = L / 2.0f, margin = 1e-4f * L;
        if (ulo + shift > -half + margin && uhi + shift < half - margin) {
            // every atom's image around z_ref is its image around the estimate:
            // est + mean MI(z - est) = est + shift + mean u
            center = gm_wrap((est + shift) + (float)(tu / (double)a.n_atoms), L, bad);
        } else {                                                      // a membrane thicker than half the box: read again
            float acc = 0.0f;
            uint32_t r2 = (dn + sh + 3u * 256u - threadIdx.x) % 3u;
            for (uint32_t c2 = threadIdx.x; c2 < n4; c2 += 256u) {
                const v4f v = src[c2];
                const uint32_t e0 = 4u * c2, j0 = r2;
                float za = v.x;
                za = j0 == 1 ? v.y : za;
                za = j0 == 2 ? v.z : za;
                if (e0 + j0 >= sh && e0 + j0 < sh + n_float) acc += gm_min_image(za - est, L, bad);
                if (j0 == 0 && e0 + 3u >= sh && e0 + 3u < sh + n_float) acc += gm_min_image(v.w - est, L, bad);
                r2 = r2 == 0 ? 2u : r2 - 1u;
            }
            const double tot = block_sum((double)acc, scratch);
            center = gm_wrap(est + (float)(tot / (double)a.n_atoms), L, bad);
        }
    }
    if (threadIdx.x == 0) {
        if (center != center || nf != 0.0 || a.n_atoms == 0) {
            raise_error(a.err, GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER, f, kStageSystem);
            center = __builtin_nanf("");
        }
        s_center = center;
    }
    __syncthreads();
    const float cdim = s_center;
    uint8_t *row = a.aflags + (size_t)(a.row0 + bi) * a.n_mol_total;
    const bool last = bi + 1 == a.n_assign;
    for (uint32_t m = threadIdx.x; m < a.n_mol_total; m += blockDim.x) {
        const float hp = x[3u * (size_t)a.heads[m] + dn];
        float d = hp - cdim;
        if (pbc) d = gm_min_image(d, L, bad);
        row[m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
        if (last && a.adist) a.adist[m] = d;
    }
    if (bad) raise_box_range(a.err, f);
}
template <bool TURNS>
__global__ __launch_bounds__(256) void k_leaflets_global_contig(LeafletArgs a) {
    if (!TURNS) {
        leaflets_global_contig_frame(a, blockIdx.x);
        return;
    }
    for (uint32_t bi = blockIdx.x; bi < a.n_assign; bi += gridDim.x) {
        leaflets_global_contig_frame(a, bi);
        __syncthreads();
    }
}

// ---- one read for global leaflets + order parameters: what follows k_bonds_tiled<..., MOM> ------------------------------
// The order kernel of a speculative batch routed every molecule by the side it had at the last assignment before the batch
// (row 0 of aflags) and left, per frame and tile, the sums of the membrane atoms' normal coordinates (FrameArgs::mom).
//   k_spec_check, first wave (k_spec_resolve's part): per frame, the plain mean c of the membrane's normal coordinate and whether it IS the reference's centre:
//                    the reference takes the circular mean as an estimate, the image of every atom next to it, and their mean
//                    (GlobalClassification, leaflets.rs:586-640 -> pbc.rs).  With angles d_j = 2 pi (z_j - c) / L about the
//                    mean (sum d_j = 0):  |sum sin d_j| = |sum (sin d_j - d_j)| <= d_max sum d_j^2 / 6  and
//                    sum cos d_j >= n - sum d_j^2 / 2,  so the estimate lies within  atan(d_max S2 / 6 / (n - S2 / 2)) L / 2 pi
//                    of c;  if every atom of the group keeps that distance, and a margin, from the far side of the box as
//                    seen from c, every image is the atom's own coordinate moved by the same number of box lengths and the
//                    centre is c (modulo L).  Frames where that cannot be shown (a membrane across the periodic boundary,
//                    a NaN) are left to k_leaflets_global_contig (`skip`).
//   k_spec_check, then: per (frame, molecule): the side by the exact centre against the side the order kernel used; the pairs
//                    (rows 1 + f of the flag table).
//   k_spec_fixup   : per pair whose exact side is not row 0's: the molecule's samples of that frame again, their ticks moved
//                    from one leaflet's sums to the other's.  Lipids do not change leaflet from one frame to the next:
//                    there are few.  k_spec_finish: the last frame's sides become the next batch's row 0.
struct SpecArgs {
    const float *xyz;
    const float *box9;
    uint32_t n_atoms, n_frames, n_tiles, n_mol_total, n_membrane;
    uint32_t dim;
    int flip, pbc;
    const float4 *mom;         // [n_frames][n_tiles]
    const float *head_z;       // [n_frames][n_mol_total] the heads' normal coordinates, from the order kernel
    float *center;             // [n_frames]
    uint8_t *ok;               // [n_frames] 1 = `center` is the reference's centre
    uint8_t *aflags;           // row 0: the sides the order kernel used; rows 1 + f: every frame's exact sides (k_spec_check's,
                               // or the exact kernel's for the frames that were not ok); k_spec_finish copies the last to row 0
    float *adist;
    uint32_t *counters;        // this batch's pair: [0] mispredicted pairs, [1] frames that were not ok
    uint32_t *counters_next;   // the next batch's pair (k_spec_fixup zeroes it)
    uint32_t *host_counters;   // pinned host copy of this batch's pair
    uint32_t *err;
};
// k_spec_check: a workgroup per frame (in turn).  Its first wave adds the frame's tiles — k_spec_resolve's part, see above —,
// then every thread compares molecules.
__global__ __launch_bounds__(256) void k_spec_check(SpecArgs a) {
    __shared__ float s_center;
    __shared__ int s_ok;
    for (uint32_t f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
        const float L = a.pbc ? a.box9[9 * (size_t)f + 4 * a.dim] : 0.0f;
        if (threadIdx.x < 64u) {
            double S = 0.0, Q = 0.0;
            float mn = 3.0e38f, mx = -3.0e38f;
            for (uint32_t t = threadIdx.x; t < a.n_tiles; t += 64u) {
                const float4 m = a.mom[(size_t)f * a.n_tiles + t];
                S += (double)m.x; Q += (double)m.y;
                mn = fminf(mn, m.z); mx = fmaxf(mx, m.w);
            }
            for (int off = 32; off >= 1; off >>= 1) {
                S += __shfl_xor(S, off, 64); Q += __shfl_xor(Q, off, 64);
                mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            }
            if (threadIdx.x == 0u) {
                const double n = (double)a.n_membrane, c = S / n;
                bool ok = (S - S == 0.0) && (Q - Q == 0.0) && mn <= mx;             // finite sums, at least one atom
                if (ok && a.pbc) {
                    const double k = 6.283185307179586 / (double)L;
                    // (the tiles' sums are f32: sum z^2 of a tile is good to ~1e-6 of itself, and the difference below is a
                    // few per cent of it in the worst case: 1e-4 of slack on the total covers it)
                    const double var = fmax(Q - S * S / n, 0.0) + 1e-4 * Q + 1e-6 * n;
                    const double dmax = k * fmax((double)mx - c, c - (double)mn), S2 = k * k * var, den = n - 0.5 * S2;
                    ok = den > 0.0 && L > 0.0f;
                    if (ok) {
                        const double dist = (dmax * S2 / 6.0 / den) / k;              // atan(x) <= x
                        const double room = 0.5 * (double)L - 1e-4 * (double)L - dist;
                        ok = (double)mx - c < room && c - (double)mn < room;
                    }
                }
                a.center[f] = (float)c;
                a.ok[f] = ok ? 1 : 0;
                s_center = (float)c;
                s_ok = ok ? 1 : 0;
                if (!ok) atomicAdd(&a.counters[1], 1u);
            }
        }
        __syncthreads();
        if (s_ok) {
            const float c = s_center;
            const bool last = f + 1 == a.n_frames;
            int bad = 0;
            for (uint32_t m = threadIdx.x; m < a.n_mol_total; m += blockDim.x) {
                float d = a.head_z[(size_t)f * a.n_mol_total + m] - c;
                if (a.pbc) d = gm_min_image(d, L, bad);
                // (row 1 + f: the frame's exact sides, as the exact kernel leaves them for the frames it decides)
                a.aflags[(size_t)(1u + f) * a.n_mol_total + m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
                if (last && a.adist) a.adist[m] = d;
            }
            if (bad) raise_box_range(a.err, f);
        }
        __syncthreads();
    }
}
// the samples of every molecule, molecule by molecule (CSR), for k_spec_fixup
struct SpecSample { uint32_t i, j, slot; };
// k_spec_fixup: a wave looks at 64 (frame, molecule) pairs at a time — rows 1 + f hold every frame's exact sides by now,
// k_spec_check's or the exact kernel's —, and for each pair whose side differs from row 0 all its lanes go over that
// molecule's samples.  No list: nothing to overflow when a batch mispredicts wholesale.
// TW: the per-frame rows (timewise.rs:130-186) hold the tick under its leaflet as well: moved there too.
template <bool ACOS_COS, bool TW>
__global__ __launch_bounds__(64) void k_spec_fixup(FrameArgs a, SpecArgs sa, const uint32_t *__restrict__ mol_begin,
                                                   const SpecSample *__restrict__ samples, unsigned long long *tw_sums,
                                                   unsigned long long *tw_cnts, uint64_t tw_row0) {
    const uint64_t n_pairs = (uint64_t)sa.n_frames * sa.n_mol_total;
    uint32_t moved = 0;
    for (uint64_t p0 = (uint64_t)blockIdx.x * 64u; p0 < n_pairs; p0 += (uint64_t)gridDim.x * 64u) {
        const uint64_t pi = p0 + threadIdx.x;
        bool differs = false;
        uint32_t exact_l = 0;
        if (pi < n_pairs) {
            exact_l = sa.aflags[(size_t)sa.n_mol_total + pi];               // row 1 + f, molecule m: offset n_mol + f n_mol + m
            differs = exact_l != sa.aflags[pi % sa.n_mol_total];
        }
        unsigned long long todo = __ballot(differs);
        moved += (uint32_t)__popcll(todo);
        while (todo) {
            const int l = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint64_t pj = p0 + (uint64_t)l;
            const uint32_t f = (uint32_t)(pj / sa.n_mol_total), m = (uint32_t)(pj - (uint64_t)f * sa.n_mol_total);
            const uint32_t exact = (uint32_t)__shfl((int)exact_l, l, 64);
            const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
            for (uint32_t q = mol_begin[m] + threadIdx.x; q < mol_begin[m + 1]; q += 64u) {
                const SpecSample sm = samples[q];
                SampleAcc acc;
                int bad = 0;
                const float *p1 = x + 3u * (size_t)sm.i, *p2 = x + 3u * (size_t)sm.j;
                bond_sample<ACOS_COS>(a, f, p1[0], p1[1], p1[2], p2[0], p2[1], p2[2], m, acc, bad);
                // Leaflet::Upper = 0: a molecule that is in the upper leaflet after all gains the tick there, one that is not
                // loses it
                unsigned long long *rep = a.rep;                        // replica 0
                if (exact == 0u) {
                    atomicAdd(&rep[(size_t)a.n_acc + sm.slot], (unsigned long long)acc.s_tot);
                    atomicAdd(&rep[3u * (size_t)a.n_acc + sm.slot], 1ull);
                } else {
                    atomicAdd(&rep[(size_t)a.n_acc + sm.slot], (unsigned long long)(-acc.s_tot));
                    atomicAdd(&rep[3u * (size_t)a.n_acc + sm.slot], ~0ull);
                }
                if (TW) {           // rows [frame][total, upper, (lower = total - upper)][slot]: the upper row gains or loses the tick
                    const size_t up = (size_t)(tw_row0 + f) * 3u * a.n_acc + (size_t)a.n_acc + sm.slot;
                    atomicAdd(&tw_sums[up], (unsigned long long)(exact == 0u ? acc.s_tot : -acc.s_tot));
                    atomicAdd(&tw_cnts[up], exact == 0u ? 1ull : ~0ull);
                }
            }
        }
    }
    if (threadIdx.x == 0u && moved) atomicAdd(&sa.counters[0], moved);
}
// the batch's housekeeping, behind the fix-up: the last frame's sides become row 0, the counters go to the host's pinned
// words, the next batch's counters start at zero
__global__ __launch_bounds__(256) void k_spec_finish(SpecArgs sa) {
    for (uint32_t m = threadIdx.x; m < sa.n_mol_total; m += blockDim.x)
        sa.aflags[m] = sa.aflags[(size_t)sa.n_frames * sa.n_mol_total + m];          // row 1 + (n_frames - 1)
    if (threadIdx.x == 0u) {
        sa.host_counters[0] = sa.counters[0];
        sa.host_counters[1] = sa.counters[1];
        sa.counters_next[0] = 0u;
        sa.counters_next[1] = 0u;
    }
}

// grid = (ceil(n_mol/256), n_assign).  IndividualClassification::identify_leaflet, leaflets.rs:777-801:
// sequential f32 sum of signed head-methyl distances along the normal.
__global__ __launch_bounds__(256) void k_leaflets_individual(LeafletArgs a) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.n_mol_total) return;
    const uint32_t f = a.aframes[blockIdx.y];
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    float L = 1.0f;
    if (a.pbc) L = a.box9[9 * (size_t)f + 4 * a.dim];
    int bad = 0;
    const float hp = x[3u * (size_t)a.heads[m] + a.dim];
    float total = 0.0f;
    for (uint32_t k = a.methyl_begin[m]; k < a.methyl_begin[m + 1]; k++) {
        const float mp = x[3u * (size_t)a.methyl_atoms[k] + a.dim];
        const float d = hp - mp;
        total += a.pbc ? gm_min_image(d, L, bad) : d;
    }
    a.aflags[(size_t)(a.row0 + blockIdx.y) * a.n_mol_total + m] =
        (uint8_t)((total >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
    if (blockIdx.y + 1 == gridDim.y && a.adist) a.adist[m] = total;
    if (bad) raise_box_range(a.err, f);
}

// ---- Local leaflets (LocalClassification, leaflets.rs:661-675 -> PBC3D::calc_local_membrane_centers,
// pbc.rs:273-318; NoPBC: pbc.rs:107-139) ---------------------------------------------------------
// For every lipid head: centre of geometry (refined Bai-Breen, like the global centre) of the membrane
// atoms whose in-plane minimum-image distance from the head is < radius (an infinite cylinder along the
// normal), then common_identify_leaflet (leaflets.rs:711-732).  The reference prunes the search with a
// CellGrid of cell edge = radius (neighbours +-1 in-plane, all cells along the normal, pbc.rs:287-292);
// here: a 2-D in-plane cell list per assignment frame, cell edge >= radius, built on the device.
//   k_local_bin     : per (slab frame, membrane atom): cell id, count
//   k_local_scan    : per slab frame: exclusive scan of the cell counts (one block)
//   k_local_scatter : per (slab frame, membrane atom): cell-ordered record (coordinates + cos/sin)
//   k_local_flags   : one wave per (slab frame, head): two passes over the 3x3 neighbour cells
// In a periodic box (the usual case; "halo" below) the pipeline is, per slab of up to 512 assignment frames:
//   k_local_sums            : per-cell fixed-point sums by LDS atomics -> the LocalEdge table, no cell list
//   k_local_decide          : a lane per head: the side from a bound on what the ring of cut cells can change, where it holds
//   -- for the frames with a head left open only (their workgroups leave at once otherwise):
//   k_local_build           : bin + scan + scatter in one kernel, a workgroup per frame (membranes of up to 65 536 atoms)
//   k_local_rowprefix       : running sums along every row of cells: LocalRowPre (exact centres), LocalEdge again
//   k_local_flags_rows_open : the frames with a head left open, every head of them: 16 lanes per head, inner cells from the
//                             sums, ring atoms one by one (k_local_flags_rows: the same for every frame, without k_local_decide)
//   k_local_flags_todo      : the general passes (k_local_flags' code) for heads the rows cannot do
constexpr uint32_t kLocalMaxCells1D = 128;
// assignment frames processed per launch group: as many as fit 4 GiB of cell-list scratch, at most 2048.  (A group costs
// seven launches and ~36 us of kernels that find nothing to do whatever its size: 512 frames a group — 1 GiB — gave 1.85 M
// frames/s on the 3 072-lipid membrane, 1 024 gave 1.97 M, 2 048 2.07 M; 256 had measured 4 % slower than 512, 128 20 %.
// The scratch is sized for every frame of a group being left open; where k_local_decide decides the frames it stays untouched.)
constexpr uint32_t kLocalSlabMax = 2048;
inline uint32_t local_slab_frames(size_t n_membrane) {
    // (the heads' to-do list, 8 bytes per head and frame, is on top: heads are a fraction of the membrane atoms)
    // (records twice — the halo copies —, the cell of every atom, two tables of cell counts, the rows' two tables of sums)
    const size_t per_frame = n_membrane * (2u * 12u + 4u) + (size_t)(2 * 4u + 16u + 16u) * kLocalMaxCells1D * (kLocalMaxCells1D + 1u) + 32u;
    const size_t n = ((size_t)4096 << 20) / per_frame;
    uint32_t cap = kLocalSlabMax;
    if (const char *ev = getenv("GORDER_HIP_LOCAL_SLAB")) cap = (uint32_t)std::max(4, atoi(ev));      // (A/B: frames per launch group)
    return (uint32_t)(n < 4 ? 4 : (n > cap ? cap : n));
}

// Two tables of running sums along each row of cells (k_local_rowprefix), an entry per cell = the cells before it in its row:
// LocalRowPre for the exact centre (sum cos, sum sin of the normal angle in f32; the normal coordinate itself in f64), and
// LocalEdge, 16 bytes = ONE load, for the bound that decides most heads without their ring (k_local_flags_rows): the cell's
// first record, sum of (z - half the box) and of its square in f32 (they only feed a bound that carries its own slack), and
// the cos / sin sums again as two 16-bit fixed-point numbers (1/16 units, modulo 2^16: a DIFFERENCE of two entries is
// exact to 1/16 as long as it is below 2048 in magnitude, i.e. the span holds fewer than 2048 records).
struct LocalRowPre { float sc, ss; double sz; };
struct alignas(16) LocalEdge { uint32_t q; float zm, sq; uint32_t cs; };
constexpr uint32_t kEdgeSpanMax = 2000;        // records in a span whose 16-bit cos / sin differences are still exact
__device__ __forceinline__ uint32_t local_edge_trig(float sc, float ss) {
    return ((uint32_t)(int)__builtin_rintf(sc * 16.0f) & 0xffffu) | ((uint32_t)(int)__builtin_rintf(ss * 16.0f) << 16);
}
__device__ __forceinline__ void local_edge_trig_diff(uint32_t hi, uint32_t lo, float &dc, float &ds) {
    dc = (float)(int16_t)((hi & 0xffffu) - (lo & 0xffffu)) * 0.0625f;
    ds = (float)(int16_t)((hi >> 16) - (lo >> 16)) * 0.0625f;
}
// a membrane atom in cell order: (in-plane a, in-plane b, normal coordinate), 12 bytes.  (Round 2 kept cos and sin of the
// normal angle next to it, 20 bytes an atom: the kernels that walk the records are bound by the bytes they pull through
// L2, and the two hardware transcendentals that make the pair again from the coordinate cost less than the 8 bytes.)
struct LocalRec { float x, y, z; };

struct LocalArgs {
    const float *xyz;
    const float *box9;
    uint32_t n_atoms;
    const uint32_t *aframes;    // [n_slab] local frame index of each assignment frame of this slab;
                                // null: the slab is the frame range frame0 .. frame0 + n_slab - 1
    uint32_t frame0;
    uint32_t n_slab;
    uint32_t row0;
    uint8_t *aflags;
    float *adist;               // written for the last frame of the whole batch only (may be null)
    int write_dist_frame;       // slab-local index whose distances go to adist (-1: none)
    uint32_t n_mol_total;
    const uint32_t *heads;
    const uint32_t *mol_slot0;  // [n_mol_total] first accumulator slot of the molecule's type (error key only; may be null)
    const uint32_t *membrane;   // null: the group is atoms 0 .. n_membrane - 1 in order (one dependent load less per atom)
    uint32_t n_membrane;
    uint4 *grid;                // [n_slab] (cells along a, cells along b, reach ka, reach kb) of each slab frame, written by
                                // k_local_scan for k_local_flags (null: not wanted)
    uint32_t dim;               // normal
    int flip, pbc;
    float radius;
    float radius_thr;           // local_radius_threshold(radius)
    // scratch, per slab frame
    uint32_t *cell_of;          // [n_slab][n_membrane]
    float *trig;                // [n_slab][rec_stride] LocalRec records in cell order (see k_local_scatter)
    uint32_t *cell_count;       // [n_slab][kLocalMaxCells1D^2 + 1] counts -> starts
    uint32_t *cell_fill;        // [n_slab][kLocalMaxCells1D^2]
    // rows of cells with a HALO (local leaflets in a periodic box): a row of the grid carries, behind its ncb cells,
    // copies of its first 2 kb cells (records included), so that the (2 kb + 1) cells a head looks at in a row are one
    // contiguous run of cells — and of records — wherever the head stands; rows are `ncb + 2 kb` cells apart
    int halo;
    int prune;                  // k_local_flags_rows: decide a head from bounds on its ring where they suffice (0: GORDER_HIP_LOCAL_NO_PRUNE)
    uint32_t rows_groups;       // k_local_flags_rows: workgroups per frame (16 heads each)
    uint32_t rec_stride;        // records per slab frame the record arrays have room for (>= n_membrane; 2 x with halo)
    LocalEdge *edge;            // [n_slab][kLocalMaxCells1D * (kLocalMaxCells1D + 1)], see LocalEdge
    LocalRowPre *rowpre;        // [n_slab][kLocalMaxCells1D * (kLocalMaxCells1D + 1)] prefix sums along each row of cells
                                // (k_local_rowprefix) for k_local_flags_rows, or null
    float4 *finfo;              // [n_slab] (min, max of the membrane's normal coordinate, 1 if every coordinate is finite, -)
    uint2 *todo;                // [1 + n_slab * n_mol_total] with agg: {count, -} then the (slab frame, head) pairs
                                // k_local_flags_rows leaves to k_local_flags_todo
    int sums;                   // the slab starts with k_local_sums: k_local_build / k_local_rowprefix then only do the frames with need[s] != 0
    uint32_t *summary;          // {frames k_local_decide left open, frames it saw} summed over the slabs of a submit (k_local_flags_todo), or null
    uint32_t *summary_host;     // the last slab of a submit: where (pinned host memory) the two sums go, or null
    uint32_t *need;             // [n_slab + 1] heads of the frame k_local_decide left open (zeroed by k_local_rowprefix; the last word: any of the slab); k_local_flags_rows
                                // then works on the frames with a count only.  null: no k_local_decide, every frame
    uint32_t *err;
};

// In-plane cell grid of one frame.  A dimension with at least 3 radii of box gets cells of radius / k
// (k = kLocalFine, less when the 128-cell cap or the box says so) and a head looks at the 2k+1 cells
// around its own; a smaller dimension is ONE cell (every atom is a candidate exactly once).  Finer cells
// cut the candidates per head from 9 r^2 (k = 1) towards the disk area pi r^2: k = 4 gives 5.1 r^2.
// The grid is this engine's own pruning device — membership itself is the exact distance test.
constexpr uint32_t kLocalFine = 4;
// with a halo (k_local_flags_rows): cells of radius / 7 — the 15 rows of cells around a head are the lanes of one DPP row
constexpr uint32_t kLocalFineRows = 7;
__device__ __forceinline__ void local_axis(float L, float radius, uint32_t &nc, uint32_t &k, uint32_t k_max = kLocalFine,
                                           uint32_t nc_max = kLocalMaxCells1D) {
    nc = 1; k = 0;
    for (uint32_t kk = k_max; kk >= 1u; kk--) {
        // cells are at least 1.0001 radius / kk wide (floor + margin), so +-kk cells reach one radius even
        // when the wrapped coordinates the cells are made from are off by a rounding error
        const float fine = floorf(L / (radius / (float)kk) * 0.9999f);
        if (fine >= (float)(2u * kk + 1u) && fine <= (float)nc_max) { nc = (uint32_t)fine; k = kk; return; }
    }
}
__device__ __forceinline__ void local_grid(const LocalArgs &a, const float *box, uint32_t &nca, uint32_t &ncb,
                                           int &da, int &db, uint32_t &ka, uint32_t &kb) {
    da = (int)((a.dim + 1u) % 3u);
    db = (int)((a.dim + 2u) % 3u);
    nca = ncb = 1;   // no periodic images to prune with: one cell holds every atom
    ka = kb = 0;
    if (a.pbc) {
        const uint32_t k_max = a.halo ? kLocalFineRows : kLocalFine;
        local_axis(box[da], a.radius, nca, ka, k_max);
        // (with the halo a row is ncb + 2 kb cells long and must still fit the tables of kLocalMaxCells1D^2 cells)
        local_axis(box[db], a.radius, ncb, kb, k_max, a.halo ? kLocalMaxCells1D - 2u * kLocalFineRows : kLocalMaxCells1D);
    }
}
// cells between the starts of two rows of the grid
__device__ __forceinline__ uint32_t local_row_stride(const LocalArgs &a, uint32_t ncb, uint32_t kb) {
    return a.halo ? ncb + 2u * kb : ncb;
}
__device__ __forceinline__ void local_grid(const LocalArgs &a, const float *box, uint32_t &nca, uint32_t &ncb,
                                           int &da, int &db) {
    uint32_t ka, kb;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
}

// ordered-integer image of a float (monotonic for every non-NaN value): atomicMin / atomicMax on floats of either sign
__device__ __forceinline__ uint32_t local_float_key(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float local_key_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// finfo[s] as the kernels keep it while it is being made: (key of the minimum, key of the maximum, 1 if a coordinate is
// not finite, -); set to (max key, 0, 0, 0) by whoever writes grid[s]
__device__ __forceinline__ void local_finfo_init(const LocalArgs &a, uint32_t s) {
    if (a.finfo) reinterpret_cast<uint4 *>(a.finfo)[s] = make_uint4(0xffffffffu, 0u, 0u, 0u);
}
__device__ __forceinline__ void frame_box(const LocalArgs &a, uint32_t f, float *box) {
    box[0] = box[1] = box[2] = 1.0f;
    if (a.pbc) {
        const float *b = a.box9 + 9 * (size_t)f;
        box[0] = b[0]; box[1] = b[4]; box[2] = b[8];
    }
}

// in-plane cell of membrane atom i in slab frame s (also stored in cell_of); rows are `ncs` cells apart
__device__ __forceinline__ uint32_t local_cell_of(const LocalArgs &a, uint32_t s, uint32_t f, uint32_t i,
                                                  const float *box, uint32_t nca, uint32_t ncb, uint32_t ncs, int da, int db) {
    const float *p = a.xyz + ((size_t)f * a.n_atoms + (a.membrane ? a.membrane[i] : i)) * 3u;
    int bad = 0;
    uint32_t ca = 0, cb = 0;
    if (a.pbc) {
        const float wa = gm_wrap(p[da], box[da], bad), wb = gm_wrap(p[db], box[db], bad);
        ca = (uint32_t)fminf(fmaxf(floorf(wa / box[da] * (float)nca), 0.0f), (float)(nca - 1u));
        cb = (uint32_t)fminf(fmaxf(floorf(wb / box[db] * (float)ncb), 0.0f), (float)(ncb - 1u));
    }
    const uint32_t c = ca * ncs + cb;
    a.cell_of[(size_t)s * a.n_membrane + i] = c;
    if (bad) raise_box_range(a.err, f);
    return c;
}

constexpr uint32_t kLocalLdsCells = 4096;   // cell counts are first aggregated per block in LDS up to this grid size

__global__ __launch_bounds__(256) void k_local_bin(LocalArgs a) {
    __shared__ uint32_t hist[kLocalLdsCells];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const uint32_t ncs = local_row_stride(a, ncb, kb), ncell = nca * ncs;
    uint32_t *count = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    const bool lds = ncell <= kLocalLdsCells;   // uniform
    if (lds) {
        for (uint32_t k = threadIdx.x; k < ncell; k += blockDim.x) hist[k] = 0;
        __syncthreads();
    }
    if (i < a.n_membrane) {
        const uint32_t c = local_cell_of(a, s, f, i, box, nca, ncb, ncs, da, db);
        if (lds) atomicAdd(&hist[c], 1u);
        else atomicAdd(&count[c], 1u);
        if (a.halo && c % ncs < 2u * kb) {          // the first 2 kb cells of a row appear again behind its last cell
            if (lds) atomicAdd(&hist[c + ncb], 1u);
            else atomicAdd(&count[c + ncb], 1u);
        }
    }
    if (lds) {   // one global atomic per cell the block touched (neighbouring atoms share cells)
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < ncell; k += blockDim.x)
            if (hist[k]) atomicAdd(&count[k], hist[k]);
    }
}

__global__ __launch_bounds__(1024) void k_local_scan(LocalArgs a) {
    __shared__ uint32_t part[1024];
    const uint32_t s = blockIdx.x;
    uint32_t *cnt = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    constexpr uint32_t N = kLocalMaxCells1D * kLocalMaxCells1D, PER = N / 1024u;
    uint32_t local[PER];
    uint32_t sum = 0;
    for (uint32_t k = 0; k < PER; k++) {
        local[k] = cnt[threadIdx.x * PER + k];
        sum += local[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < PER; k++) {
        cnt[threadIdx.x * PER + k] = run;
        run += local[k];
    }
    if (threadIdx.x == 1023) cnt[N] = run;
    if (threadIdx.x == 0 && a.grid) {     // the frame's cell grid, once per frame instead of once per head
        float box[3];
        frame_box(a, a.aframes ? a.aframes[s] : a.frame0 + s, box);
        uint32_t nca, ncb, ka, kb;
        int da, db;
        local_grid(a, box, nca, ncb, da, db, ka, kb);
        a.grid[s] = make_uint4(nca, ncb, ka, kb);
        local_finfo_init(a, s);
    }
}

// Places every membrane atom in its cell's run and writes a cell-ordered RECORD next to it so that the
// flags kernel streams contiguous data instead of chasing two indices per candidate:
//   rec[q] = (in-plane a, in-plane b, normal coordinate)   (LocalRec)
__global__ __launch_bounds__(256) void k_local_scatter(LocalArgs a) {
    __shared__ uint32_t hist[kLocalLdsCells];   // per-block count, then the block's base offset in each cell
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const uint32_t ncs = local_row_stride(a, ncb, kb), ncell = nca * ncs;
    const int dn = (int)a.dim;
    const bool lds = ncell <= kLocalLdsCells;   // uniform
    uint32_t *fill = a.cell_fill + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D);
    const bool valid = i < a.n_membrane;
    uint32_t c = 0, rank = 0;
    if (valid) c = a.cell_of[(size_t)s * a.n_membrane + i];
    if (lds) {
        for (uint32_t k = threadIdx.x; k < ncell; k += blockDim.x) hist[k] = 0;
        __syncthreads();
        if (valid) rank = atomicAdd(&hist[c], 1u);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < ncell; k += blockDim.x)
            if (hist[k]) hist[k] = atomicAdd(&fill[k], hist[k]);   // reserve the block's run inside the cell
        __syncthreads();
        if (valid) rank += hist[c];
    } else if (valid) {
        rank = atomicAdd(&fill[c], 1u);
    }
    if (!valid) return;
    const uint32_t start = a.cell_count[(size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u) + c];
    const float *p = a.xyz + ((size_t)f * a.n_atoms + (a.membrane ? a.membrane[i] : i)) * 3u;
    const size_t q = (size_t)s * a.rec_stride + start + rank;
    reinterpret_cast<LocalRec *>(a.trig)[q] = LocalRec{p[da], p[db], p[dn]};
    if (a.halo && c % ncs < 2u * kb) {              // the copy in the row's halo
        const uint32_t c2 = c + ncb;
        const size_t q2 = (size_t)s * a.rec_stride + a.cell_count[(size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u) + c2] +
                          atomicAdd(&fill[c2], 1u);
        reinterpret_cast<LocalRec *>(a.trig)[q2] = LocalRec{p[da], p[db], p[dn]};
    }
}

template <int CTRL>
__device__ __forceinline__ float row_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ float row_shifted(float v) {      // out-of-row lanes keep their own value
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double row_add_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ uint32_t row_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// the value one lane holds, for every lane (lane index known at compile time: v_readlane, no LDS)
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_value(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// Per row of cells: running sums along the row (the halo cells continue it), so that k_local_flags_rows takes the
// cells of a row that lie wholly inside a head's cylinder — always one span of consecutive cells — as the difference
// of two entries:  rowpre[ra (ncs + 1) + j] = sums over the cells 0 .. j-1 of row ra (cos and sin of the normal angle
// in f32: they only make the estimate that anchors the image choice; the normal coordinate itself in f64).
// (k_local_rowprefix, below: a scan over the row's RECORDS, read off at the cells' first records.)  The same pass makes the frame's record for the rows kernel: finfo[s] = (min, max of the normal
// coordinate over the membrane, whether a coordinate of a record is not finite), kept as ordered integers.
// block = 256 threads = 4 waves = 4 rows of cells, a wave per row; grid = (ceil(kLocalMaxCells1D / 4), n_slab).
// The wave streams its row's records (contiguous: the row's cells one after the other, halo copies included) 384 at a
// time with coalesced loads — all six loads of a piece in flight together —, scans them (inclusive scan over the 64
// lanes by row shifts + the totals of the 16-lane rows before, carried from load to load), parks the exclusive sums in
// LDS, and then gives every cell of the row the sum parked at its first record.  (A lane per CELL summing its own
// handful of records took 140 us per 256 frames: 64 short gathers per load instruction.)
// One row of cells by one wave: PIECE records at a time (PIECE / 64 coalesced loads in flight), pc / ps / pz = the
// wave's PIECE + 1 parking places in LDS.  zlo / zhi / nf collect the frame's extrema and non-finite flag.
template <uint32_t PIECE>
__device__ __forceinline__ void local_row_prefix(const uint32_t *__restrict__ cstart, const LocalRec *__restrict__ rec,
                                                 LocalRowPre *__restrict__ out, LocalEdge *__restrict__ out_edge, uint32_t ra,
                                                 uint32_t ncs, float inv_Ln,
                                                 float z_mid, float *pc, float *ps, double *pz, float *pq, float &zlo, float &zhi,
                                                 uint32_t &nf) {
    const uint32_t lane = threadIdx.x & 63u;
    // the row's cell starts first, all of them at once (cell j = lane + 64 i; entry ncs = the end of the row): the record
    // loads below then depend on ONE round trip, and the cells' read-out at the end of a piece on none
    constexpr uint32_t CQ = kLocalMaxCells1D / 64u + 1u;
    uint32_t cq[CQ];
#pragma unroll
    for (uint32_t i = 0; i < CQ; i++) cq[i] = lane + 64u * i <= ncs ? cstart[ra * ncs + lane + 64u * i] : 0xffffffffu;
    const uint32_t qa = (uint32_t)__builtin_amdgcn_readfirstlane((int)cq[0]);
    uint32_t qb = 0;
#pragma unroll
    for (uint32_t i = 0; i < CQ; i++)
        if ((ncs >> 6) == i) qb = (uint32_t)__builtin_amdgcn_readlane((int)cq[i], (int)(ncs & 63u));       // (uniform)
    float carry_c = 0.0f, carry_s = 0.0f, carry_q = 0.0f;
    double carry_z = 0.0;
    static_assert(PIECE == 256u, "a lane takes four consecutive records of a piece");
    for (uint32_t base = qa; base < qb || base == qa; base += PIECE) {
        const uint32_t n_here = min(PIECE, qb - base);
        // FOUR CONSECUTIVE records per lane (the wave's four loads walk the same cache lines): the running sums inside a
        // lane are three adds per quantity, and the scan over the 64 lanes — row shifts, the rows' totals by v_readlane —
        // runs once per piece on the lanes' totals instead of once per 64 records (the scan was 60 % of this kernel's
        // instructions: 128 -> see DESIGN K6)
        LocalRec r[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t q = base + 4u * lane + k;
            r[k] = rec[q < qb ? q : (qb ? qb - 1u : 0u)];
        }
        float vc[4], vs[4], vq[4];
        double vz[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const bool valid = base + 4u * lane + k < qb;
            float sn_t, cs_t;
            local_trig(r[k].z, inv_Ln, &sn_t, &cs_t);
            vc[k] = valid ? cs_t : 0.0f;
            vs[k] = valid ? sn_t : 0.0f;
            vz[k] = valid ? (double)r[k].z : 0.0;
            const float dq = r[k].z - z_mid;
            vq[k] = valid ? dq * dq : 0.0f;
            if (valid) {
                zlo = fminf(zlo, r[k].z);
                zhi = fmaxf(zhi, r[k].z);
                nf |= ((r[k].x - r[k].x) + (r[k].y - r[k].y)) + (r[k].z - r[k].z) == 0.0f ? 0u : 1u;
            }
        }
        // the lane's own running sums: x1 = v0, x2 = v0 + v1, x3 = (v0 + v1) + v2, total = x3 + v3
        const float c1 = vc[0], c2 = c1 + vc[1], c3 = c2 + vc[2], tc = c3 + vc[3];
        const float s1 = vs[0], s2 = s1 + vs[1], s3 = s2 + vs[2], ts = s3 + vs[3];
        const float q1 = vq[0], q2 = q1 + vq[1], q3 = q2 + vq[2], tq = q3 + vq[3];
        const double z1 = vz[0], z2 = z1 + vz[1], z3 = z2 + vz[2], tz = z3 + vz[3];
        float ic = tc, is = ts, iq = tq;
        double iz = tz;
        ic = row_add<0x111>(ic); is = row_add<0x111>(is); iz = row_add_f64<0x111>(iz); iq = row_add<0x111>(iq);
        ic = row_add<0x112>(ic); is = row_add<0x112>(is); iz = row_add_f64<0x112>(iz); iq = row_add<0x112>(iq);
        ic = row_add<0x114>(ic); is = row_add<0x114>(is); iz = row_add_f64<0x114>(iz); iq = row_add<0x114>(iq);
        ic = row_add<0x118>(ic); is = row_add<0x118>(is); iz = row_add_f64<0x118>(iz); iq = row_add<0x118>(iq);
        float bc = 0.0f, bs = 0.0f, bq = 0.0f;
        double bz = 0.0;
#pragma unroll
        for (int r4 = 0; r4 < 3; r4++) {              // (the rows' totals by v_readlane: a __shfl is a trip through the LDS each)
            const float rc = lane_value(ic, 16 * r4 + 15), rs = lane_value(is, 16 * r4 + 15), rq = lane_value(iq, 16 * r4 + 15);
            const double rz = lane_value(iz, 16 * r4 + 15);
            if ((int)(lane >> 4) > r4) { bc += rc; bs += rs; bz += rz; bq += rq; }
        }
        ic += bc; is += bs; iz += bz; iq += bq;                             // inclusive over the 64 lanes' totals
        const float ec = carry_c + (ic - tc), es = carry_s + (is - ts), eq = carry_q + (iq - tq);   // records of the row before this lane's
        const double ez = carry_z + (iz - tz);
        *reinterpret_cast<float4 *>(pc + 4u * lane) = make_float4(ec, ec + c1, ec + c2, ec + c3);
        *reinterpret_cast<float4 *>(ps + 4u * lane) = make_float4(es, es + s1, es + s2, es + s3);
        *reinterpret_cast<float4 *>(pq + 4u * lane) = make_float4(eq, eq + q1, eq + q2, eq + q3);
        *reinterpret_cast<double2 *>(pz + 4u * lane) = make_double2(ez, ez + z1);
        *reinterpret_cast<double2 *>(pz + 4u * lane + 2u) = make_double2(ez + z2, ez + z3);
        carry_c += lane_value(ic, 63);
        carry_s += lane_value(is, 63);
        carry_z += lane_value(iz, 63);
        carry_q += lane_value(iq, 63);
        if (lane == 0u) { pc[n_here] = carry_c; ps[n_here] = carry_s; pz[n_here] = carry_z; pq[n_here] = carry_q; }   // behind the piece's last record
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the cells whose first record lies in this piece (a cell that starts where the piece ends belongs to the next
        // piece — or, behind the row's last record, to this one)
        const bool last = base + n_here >= qb;
#pragma unroll
        for (uint32_t i = 0; i < CQ; i++) {
            const uint32_t j = lane + 64u * i, q0 = cq[i];
            if (j <= ncs && q0 >= base && (q0 < base + n_here || (last && q0 == qb))) {
                LocalRowPre o;
                o.sc = pc[q0 - base]; o.ss = ps[q0 - base]; o.sz = pz[q0 - base];
                out[j] = o;
                LocalEdge e;
                e.q = q0;
                e.zm = (float)(o.sz - (double)(q0 - qa) * (double)z_mid);
                e.sq = pq[q0 - base];
                e.cs = local_edge_trig(o.sc, o.ss);
                out_edge[j] = e;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (qb == qa) break;                                                // (an empty row: one trip for its cells)
    }
}
// the frame's extrema of the normal coordinate and its "a coordinate is not finite" flag, from one wave
__device__ __forceinline__ void local_finfo_merge(const LocalArgs &a, uint32_t s, float zlo, float zhi, uint32_t nf) {
    for (int off = 32; off >= 1; off >>= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, off, 64));
        zhi = fmaxf(zhi, __shfl_xor(zhi, off, 64));
        nf |= (uint32_t)__shfl_xor((int)nf, off, 64);
    }
    if ((threadIdx.x & 63u) == 0u) {
        uint32_t *fi = reinterpret_cast<uint32_t *>(a.finfo + s);
        if (zlo <= zhi) {
            atomicMin(fi, local_float_key(zlo));
            atomicMax(fi + 1, local_float_key(zhi));
        }
        if (nf) atomicOr(fi + 2, 1u);
    }
}
// (pieces of 128 to 640 records — 8 to 41 KB of LDS per workgroup — all take 87-89 us per 256 frames of the 3072-lipid
// membrane: neither the waves in flight nor the round trips per row bound this kernel; 256 keeps the LDS small)
constexpr uint32_t kRowPrefixPiece = 256;
__global__ __launch_bounds__(256) void k_local_rowprefix(LocalArgs a) {
    // (+ 1 for the sums behind a full piece's last record; rows 16 bytes apart in size: the lanes store four values at once)
    __shared__ alignas(16) float l_c[4][kRowPrefixPiece + 4], l_s[4][kRowPrefixPiece + 4], l_q[4][kRowPrefixPiece + 4];
    __shared__ alignas(16) double l_z[4][kRowPrefixPiece + 2];
    const uint32_t s = blockIdx.y, wave = threadIdx.x >> 6, ra = blockIdx.x * 4u + wave;
    const uint4 g = a.grid[s];
    const uint32_t nca = g.x, ncs = local_row_stride(a, g.y, g.w);
    if (a.sums) {                                   // behind k_local_sums + k_local_decide: the frames with a head left open only
        if (a.need[a.n_slab] == 0u || a.need[s] == 0u) return;
    } else if (blockIdx.x == 0 && s == 0 && threadIdx.x == 0) a.todo[0] = make_uint2(0u, 0u);   // this slab's list starts empty
    if (!a.sums && blockIdx.x == 0 && threadIdx.x == 0 && a.need) {                              // nothing of this frame left open yet
        a.need[s] = 0u;
        if (s == 0) a.need[a.n_slab] = 0u;                                                      // (nor of the slab: the word behind the frames')
    }
    if (ra >= nca) return;                                                                       // (uniform per wave; no workgroup barrier below)
    const bool merge = reinterpret_cast<const uint4 *>(a.finfo)[s].w != 2u;                     // 2: k_local_build made the frame's record
    const uint32_t *cstart = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    const LocalRec *rec = reinterpret_cast<const LocalRec *>(a.trig) + (size_t)s * a.rec_stride;
    float box[3];
    frame_box(a, a.aframes ? a.aframes[s] : a.frame0 + s, box);
    LocalRowPre *out = a.rowpre + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u)) + (size_t)ra * (ncs + 1u);
    LocalEdge *out_edge = a.edge + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u)) + (size_t)ra * (ncs + 1u);
    float zlo = 3.0e38f, zhi = -3.0e38f;
    uint32_t nf = 0;
    local_row_prefix<kRowPrefixPiece>(cstart, rec, out, out_edge, ra, ncs, 1.0f / box[a.dim], 0.5f * box[a.dim], l_c[wave], l_s[wave], l_z[wave],
                                      l_q[wave], zlo, zhi, nf);
    if (merge) local_finfo_merge(a, s, zlo, zhi, nf);
}

// bin + scan + scatter in ONE kernel, one block per slab frame (membranes of up to kLocalBuildMax atoms: the slab then has
// enough frames to fill the chip with one block each).  The cell counts live in LDS (no global atomics, no memsets,
// no cell_of round trip): pass 1 counts — the counting atomic's return value is the atom's place inside its cell, which
// the thread keeps (a byte per atom) —, an in-block scan turns the counts into the starts (kept in LDS and written to
// cell_count for the kernels that follow), pass 2 recomputes an atom's cell — the same arithmetic on the same
// coordinates, now an L2 hit — and writes the record at start + place.  LDS atomics are served one LANE per cycle on
// this hardware (PMC: the LDS busy 56 % of a version with five atomics per atom, 64 cycles per instruction), so their
// number is what the kernel costs: one per atom and one per halo copy here.  A cell with more than 255 atoms (a place
// that does not fit its byte) makes pass 2 take the places from a second LDS counter instead.
// dynamic LDS: 2 x kLocalMaxCells1D^2 words.
constexpr uint32_t kLocalBuildMax = 65536;
constexpr uint32_t kLocalBuildLds = 2u * kLocalMaxCells1D * kLocalMaxCells1D * (uint32_t)sizeof(uint32_t);
constexpr uint32_t kLocalBuildTrips = kLocalBuildMax / (8u * 1024u);       // trips of eight atoms a thread makes at most
// TRIPS: trips of eight atoms a thread makes at most (2, 5 or kLocalBuildTrips: the places it keeps cost registers)
template <uint32_t TRIPS>
__global__ __launch_bounds__(1024) void k_local_build(LocalArgs a) {
    extern __shared__ uint32_t l_build[];
    __shared__ uint32_t l_wave[16];
    __shared__ uint32_t l_over;
    uint32_t *l_start = l_build, *l_fill = l_build + kLocalMaxCells1D * kLocalMaxCells1D;
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    if (a.sums && (a.need[a.n_slab] == 0u || a.need[s] == 0u)) return;      // behind k_local_sums + k_local_decide: open frames only
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const uint32_t ncs = local_row_stride(a, ncb, kb), ncell = nca * ncs;
    const uint32_t n_halo = a.halo ? 2u * kb : 0u;     // columns of a row that appear again behind its last cell
    const int dn = (int)a.dim;
    for (uint32_t k = tid; k < ncell; k += 1024u) { l_start[k] = 0; l_fill[k] = 0; }
    if (tid == 0) l_over = 0;
    __syncthreads();
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    int bad = 0;
    bool in_halo = false;
    // cells per length, once (an atom within an ulp or two of a cell edge may land in the neighbour of the cell the
    // division wa / L * n would give: the consumers' margins — 4e-4 r^2 and 1e-3 of a cell in the rows kernel, cells
    // 1.0001 r / k wide in the general passes — are orders of magnitude wider; membership itself is always the exact test)
    const float inv_a = a.pbc ? (float)nca / box[da] : 0.0f, inv_b = a.pbc ? (float)ncb / box[db] : 0.0f;
    auto cell_ab = [&](float xa, float xb) -> uint32_t {
        in_halo = false;
        if (!a.pbc) return 0u;
        // gm_wrap's first steps as selects — the same subtraction and addition —; a coordinate more than a box length
        // outside (rare) takes the loops.  (Four data-dependent loops per atom, forty atoms per thread, were most of this
        // kernel's branches: a taken branch costs a wave far more than the arithmetic it skips.)
        const float La = box[da], Lb = box[db];
        float wa = xa > La ? xa - La : xa, wb = xb > Lb ? xb - Lb : xb;
        wa = wa < 0.0f ? wa + La : wa;
        wb = wb < 0.0f ? wb + Lb : wb;
        if (__builtin_expect(!(wa >= 0.0f && wa <= La && wb >= 0.0f && wb <= Lb), 0)) {      // (NaN comes here too, and stays NaN)
            wa = gm_wrap(xa, La, bad);
            wb = gm_wrap(xb, Lb, bad);
        }
        const uint32_t ca = (uint32_t)fminf(fmaxf(floorf(wa * inv_a), 0.0f), (float)(nca - 1u));
        const uint32_t cb = (uint32_t)fminf(fmaxf(floorf(wb * inv_b), 0.0f), (float)(ncb - 1u));
        in_halo = cb < n_halo;
        return ca * ncs + cb;
    };
    // eight atoms per trip: the index and coordinate loads of all of them go out before the first is used (a thread walks
    // ~36 atoms; one dependent load pair per atom would leave the 16 waves of the block waiting most of the time)
    constexpr uint32_t U = 8;
    uint32_t place[TRIPS][2], place2[TRIPS][2];       // a byte per atom: its place in its cell / in the halo cell
    uint32_t cells[TRIPS][4];                         // 16 bits per atom: its cell (14 bits) and the "has a halo copy" bit, for pass 2
    bool over = false;
#pragma unroll
    for (uint32_t trip = 0; trip < TRIPS; trip++) {
        const uint32_t i0 = tid + trip * U * 1024u;
        place[trip][0] = place[trip][1] = place2[trip][0] = place2[trip][1] = 0u;
        cells[trip][0] = cells[trip][1] = cells[trip][2] = cells[trip][3] = 0u;
        if (trip * U * 1024u >= a.n_membrane) continue;                     // (uniform)
        uint32_t at[U];
        float pa[U], pb[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t i = min(i0 + u * 1024u, a.n_membrane - 1u); at[u] = a.membrane ? a.membrane[i] : i; }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) { pa[u] = x[3u * (size_t)at[u] + da]; pb[u] = x[3u * (size_t)at[u] + db]; }
#pragma unroll
        for (uint32_t u = 0; u < U; u++)
            if (i0 + u * 1024u < a.n_membrane) {
                const uint32_t c = cell_ab(pa[u], pb[u]);
                cells[trip][u >> 1] |= (c | (in_halo ? 0x4000u : 0u)) << (16u * (u & 1u));
                const uint32_t r = atomicAdd(&l_start[c], 1u);
                over |= r > 255u;
                place[trip][u >> 2] |= (r & 255u) << (8u * (u & 3u));
                if (in_halo) {
                    const uint32_t r2 = atomicAdd(&l_start[c + ncb], 1u);
                    over |= r2 > 255u;
                    place2[trip][u >> 2] |= (r2 & 255u) << (8u * (u & 3u));
                }
            }
    }
    if (over) l_over = 1u;
    __syncthreads();
    const bool use_fill = l_over != 0u;                 // (uniform) some cell holds more atoms than a byte counts
    // exclusive scan: 16 consecutive cells per thread, wave scan of the thread sums, then the waves' totals
    {
        constexpr uint32_t PER = kLocalMaxCells1D * kLocalMaxCells1D / 1024u;
        const uint32_t c0 = tid * PER;
        uint32_t cnt[PER], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            cnt[k] = c0 + k < ncell ? l_start[c0 + k] : 0u;
            sum += cnt[k];
        }
        uint32_t incl = sum;
#pragma unroll
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if ((tid & 63u) >= off) incl += v;
        }
        if ((tid & 63u) == 63u) l_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < (tid >> 6); w++) run += l_wave[w];
        uint32_t *cstart = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) {
            if (c0 + k <= ncell) {              // (entry ncell = the number of atoms)
                cstart[c0 + k] = run;
                if (c0 + k < ncell) l_start[c0 + k] = run;
            }
            run += cnt[k];
        }
        // a grid that fills the table (128 rows of 114 + 2 * 7 cells): entry ncell lies behind the last thread's cells
        if (tid == 1023u && c0 + PER == ncell) cstart[ncell] = run;
    }
    if (tid == 0 && a.grid) { a.grid[s] = make_uint4(nca, ncb, ka, kb); local_finfo_init(a, s); }
    __syncthreads();
    LocalRec *rec = reinterpret_cast<LocalRec *>(a.trig) + (size_t)s * a.rec_stride;
    float zlo = 3.0e38f, zhi = -3.0e38f;            // the frame's record for the rows kernel (finfo), made on the way
    uint32_t nf = 0;
#pragma unroll
    for (uint32_t trip = 0; trip < TRIPS; trip++) {
        const uint32_t i0 = tid + trip * U * 1024u;
        if (trip * U * 1024u >= a.n_membrane) continue;                     // (uniform)
        uint32_t at[U];
        float pa[U], pb[U], pn[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t i = min(i0 + u * 1024u, a.n_membrane - 1u); at[u] = a.membrane ? a.membrane[i] : i; }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            pa[u] = x[3u * (size_t)at[u] + da]; pb[u] = x[3u * (size_t)at[u] + db]; pn[u] = x[3u * (size_t)at[u] + dn];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (i0 + u * 1024u >= a.n_membrane) continue;
            const uint32_t cw = (cells[trip][u >> 1] >> (16u * (u & 1u))) & 0xffffu;       // (pass 1's cell: not computed twice)
            const uint32_t c = cw & 0x3fffu;
            in_halo = (cw & 0x4000u) != 0u;
            const uint32_t q = l_start[c] + (use_fill ? atomicAdd(&l_fill[c], 1u) : (place[trip][u >> 2] >> (8u * (u & 3u))) & 255u);
            rec[q] = LocalRec{pa[u], pb[u], pn[u]};
            zlo = fminf(zlo, pn[u]);
            zhi = fmaxf(zhi, pn[u]);
            nf |= ((pa[u] - pa[u]) + (pb[u] - pb[u])) + (pn[u] - pn[u]) == 0.0f ? 0u : 1u;
            if (in_halo) {
                const uint32_t q2 = l_start[c + ncb] + (use_fill ? atomicAdd(&l_fill[c + ncb], 1u) : (place2[trip][u >> 2] >> (8u * (u & 3u))) & 255u);
                rec[q2] = LocalRec{pa[u], pb[u], pn[u]};
            }
        }
    }
    if (bad) raise_box_range(a.err, f);
    // The extrema of the normal coordinate and the "a coordinate is not finite" flag of the frame, in one place: 84 waves
    // of k_local_rowprefix adding theirs by atomics — 2 000 of them per cache line of finfo — was what that kernel took
    // most of its time for.
    if (a.finfo) {
        __shared__ float l_zlo[16], l_zhi[16];
        __shared__ uint32_t l_nf[16];
        for (int off = 32; off >= 1; off >>= 1) {
            zlo = fminf(zlo, __shfl_xor(zlo, off, 64));
            zhi = fmaxf(zhi, __shfl_xor(zhi, off, 64));
            nf |= (uint32_t)__shfl_xor((int)nf, off, 64);
        }
        if ((tid & 63u) == 0u) { l_zlo[tid >> 6] = zlo; l_zhi[tid >> 6] = zhi; l_nf[tid >> 6] = nf; }
        __syncthreads();
        if (tid == 0) {
            for (uint32_t w = 1; w < 16u; w++) { zlo = fminf(zlo, l_zlo[w]); zhi = fmaxf(zhi, l_zhi[w]); nf |= l_nf[w]; }
            reinterpret_cast<uint4 *>(a.finfo)[s] = zlo <= zhi ? make_uint4(local_float_key(zlo), local_float_key(zhi), nf, 2u)
                                                                : make_uint4(0xffffffffu, 0u, nf, 2u);
        }
    }
    // (The rows' prefix sums stay a kernel of their own, k_local_rowprefix: done here behind pass 2 — a wave per row, six
    // rows per wave, the records still in this XCD's L2 — they took 190 us per 256 frames against 100 for the kernel,
    // whose 23 000 waves hide the latency of the four dependent loads a row needs; with per-cell sums made by LDS
    // atomics in pass 2 instead, +130 us: LDS atomics are served one lane per cycle.)
}

// `sqrt(d2) < radius` (groan_rs Cylinder::inside) is evaluated as `d2 < thr` with thr = the smallest float
// whose correctly rounded square root reaches the radius: sqrt is monotonic, so the two tests select
// exactly the same atoms.  Computed once on the host; k_local_flags gets it as LocalArgs::radius_thr.
__host__ __device__ inline float local_radius_threshold(float r) {
    if (!(r > 0.0f)) return 0.0f;            // sqrt(x) < r never holds
    float thr = r * r;
    for (int i = 0; i < 8 && sqrtf(thr) < r; i++) thr = nextafterf(thr, INFINITY);
    for (int i = 0; i < 8; i++) {
        const float p = nextafterf(thr, 0.0f);
        if (!(p < thr) || !(sqrtf(p) >= r)) break;
        thr = p;
    }
    return thr;
}

// Sum over the 64 lanes by DPP row shifts (cheaper than six ds_bpermute round trips per sum).
// Within a row of 16 lanes a Hillis-Steele scan leaves the row total in its last lane; row_bcast:15 and
// row_bcast:31 carry the totals on, lane 63 ends with the wave total.  Fixed order => deterministic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_total(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);   // row_shr:8
    v = dpp_add_f64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}


// The same row-shift reductions for values whose last digits do not reach the result (the circular sums only make the
// ESTIMATE, the member count is an integer): one DPP-modified add / min / max per step instead of three for an f64.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_f32(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_total_f32(float v) {
    v = dpp_add_f32<0x111, 0xf>(v); v = dpp_add_f32<0x112, 0xf>(v); v = dpp_add_f32<0x114, 0xf>(v); v = dpp_add_f32<0x118, 0xf>(v);
    v = dpp_add_f32<0x142, 0xa>(v); v = dpp_add_f32<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {
    v = dpp_add_u32<0x111, 0xf>(v); v = dpp_add_u32<0x112, 0xf>(v); v = dpp_add_u32<0x114, 0xf>(v); v = dpp_add_u32<0x118, 0xf>(v);
    v = dpp_add_u32<0x142, 0xa>(v); v = dpp_add_u32<0x143, 0xc>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// minimum / maximum over the wave: lanes without a source keep their own value (bound_ctrl off, old = the value itself)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f32(float v) {
    v = fminf(v, dpp_keep_f32<0x111, 0xf>(v)); v = fminf(v, dpp_keep_f32<0x112, 0xf>(v));
    v = fminf(v, dpp_keep_f32<0x114, 0xf>(v)); v = fminf(v, dpp_keep_f32<0x118, 0xf>(v));
    v = fminf(v, dpp_keep_f32<0x142, 0xa>(v)); v = fminf(v, dpp_keep_f32<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_keep_f32<0x111, 0xf>(v)); v = fmaxf(v, dpp_keep_f32<0x112, 0xf>(v));
    v = fmaxf(v, dpp_keep_f32<0x114, 0xf>(v)); v = fmaxf(v, dpp_keep_f32<0x118, 0xf>(v));
    v = fmaxf(v, dpp_keep_f32<0x142, 0xa>(v)); v = fmaxf(v, dpp_keep_f32<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// block = 256 threads = 4 waves = 4 heads; grid = (ceil(n_mol / 4), n_slab).  A head's candidates are
// the records of the (2ka+1) x (2kb+1) cells around its own: per row of cells ONE contiguous run of
// records (two when the run wraps around the box), lanes over the run.
// The general passes for ONE head by ONE wave (all 64 lanes must be here).
__device__ __noinline__ void local_flags_head(const LocalArgs &a, uint32_t s, uint32_t m) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    if (a.grid) {         // uniform per workgroup: scalar loads
        const uint4 g = a.grid[s];
        nca = g.x; ncb = g.y; ka = g.z; kb = g.w;
        da = (int)((a.dim + 1u) % 3u);
        db = (int)((a.dim + 2u) % 3u);
    } else {
        local_grid(a, box, nca, ncb, da, db, ka, kb);
    }
    const int dn = (int)a.dim;
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    const float *hp = x + 3u * (size_t)a.heads[m];
    const float ha_pos = hp[da], hb_pos = hp[db], hn_pos = hp[dn];
    int bad = 0;
    uint32_t ha = 0, hb = 0;
    if (a.pbc) {
        const float wa = gm_wrap(ha_pos, box[da], bad), wb = gm_wrap(hb_pos, box[db], bad);
        ha = (uint32_t)fminf(fmaxf(floorf(wa / box[da] * (float)nca), 0.0f), (float)(nca - 1u));
        hb = (uint32_t)fminf(fmaxf(floorf(wb / box[db] * (float)ncb), 0.0f), (float)(ncb - 1u));
    }
    const uint32_t *cstart = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    const LocalRec *rec = reinterpret_cast<const LocalRec *>(a.trig) + (size_t)s * a.rec_stride;
    const uint32_t ncs = local_row_stride(a, ncb, kb);      // cells between two rows of the grid
    const float La = box[da], Lb = box[db], Ln = box[dn], inv_Ln = 1.0f / Ln;
    const float thr = a.radius_thr;
    // rows (ha - ka .. ha + ka) mod nca; in a row the cells (hb - kb .. hb + kb) mod ncb = runs [b0, b1) and [0, b2)
    const uint32_t n_rows = 2u * ka + 1u, n_cols = 2u * kb + 1u;       // <= nca, ncb by local_axis
    // (x + n - k) mod n for x < n, k <= n: one conditional subtraction (an integer modulo is ~15 instructions)
    uint32_t a0 = ha + nca - ka, b0 = hb + ncb - kb;
    a0 -= a0 >= nca ? nca : 0u;
    b0 -= b0 >= ncb ? ncb : 0u;
    // (with the halo the cells b0 .. b0 + n_cols - 1 of a row are there as such: one run, never a second one)
    const uint32_t b1 = a.halo ? b0 + n_cols : min(b0 + n_cols, ncb), b2 = b0 + n_cols - b1;
    const bool pbc = a.pbc != 0;
    auto inside = [&](float ra, float rb) {
        float ea = ra - ha_pos, eb = rb - hb_pos;
        if (pbc) {
            bool slow = false;
            const float fa = gm_min_image_step(ea, La, slow), fb = gm_min_image_step(eb, Lb, slow);
            if (__builtin_expect(slow, 0)) {
                ea = gm_min_image_loop(ea, La, bad);
                eb = gm_min_image_loop(eb, Lb, bad);
            } else {
                ea = fa; eb = fb;
            }
        }
        return ea * ea + eb * eb < thr;                 // == sqrt(..) < radius, see local_radius_threshold
    };

    // The runs as a flat list of wave iterations: lane i keeps (first record, end of run) of iteration i.
    // Every load address of the passes below then comes from a lane read-out instead of a chain of
    // dependent cell-table loads, so the loads of several iterations are in flight together — this
    // kernel is bound by load latency, not by arithmetic.
    float sc = 0.0f, ss = 0.0f, sp = 0.0f, su = 0.0f, ulo = 3.0e38f, uhi = -3.0e38f;
    uint32_t cnt = 0, nf = 0;
    bool done = false;
    const uint32_t n_runs = 2u * n_rows;
    uint32_t rq0 = 0, rq1 = 0;
    if (!done && lane < n_runs) {       // (the run table serves the general passes only)
        uint32_t ra = a0 + (lane >> 1);
        ra -= ra >= nca ? nca : 0u;
        const uint32_t row = ra * ncs;
        rq0 = (lane & 1u) ? cstart[row] : cstart[row + b0];
        rq1 = (lane & 1u) ? cstart[row + b2] : cstart[row + b1];
    }
    // Every run overwrites the lanes from its first iteration on (a later run then overwrites its own); the lanes
    // past the last iteration are cleared at the end.  The second run of a row is empty unless the columns wrap.
    uint32_t n_it = 0, it_base = 0, it_end = 0;
    for (uint32_t r = 0; !done && r < n_runs; r += (b2 ? 1u : 2u)) {
        const uint32_t q0 = __builtin_amdgcn_readlane(rq0, r), q1 = __builtin_amdgcn_readlane(rq1, r);
        const bool from_here = lane >= n_it;
        it_base = from_here ? q0 + 64u * (lane - n_it) : it_base;
        it_end = from_here ? q1 : it_end;
        n_it += (q1 - q0 + 63u) >> 6;
    }
    if (lane >= n_it) { it_base = 0; it_end = 0; }
    const bool flat = n_it <= 64u;       // else (> 4096 candidates): the plain run loops

    // pass 1: members (in-plane minimum-image distance < radius; groan_rs Cylinder::inside), their count
    // and the circular sums of the normal coordinate (PBC) or its plain sum (NoPBC).
    // The refinement sum of pass 2 is taken here already, relative to the head's own normal coordinate:
    // u = MI(z - z_head), with its minimum and maximum over the members.  If, once the estimate is known, every
    // member's image around the head is also its image around the estimate, then sum MI(z - est) = sum u +
    // n MI(z_head - est) and pass 2 is not needed (k_leaflets_global_contig explains the argument).
    auto take = [&](const LocalRec r) {
        if (inside(r.x, r.y)) {
            cnt += 1;
            nf |= (r.z - r.z == 0.0f) ? 0u : 1u;
            if (pbc) {
                float sn, cs;
                local_trig(r.z, inv_Ln, &sn, &cs);
                sc += cs; ss += sn;
                const float u = gm_min_image(r.z - hn_pos, Ln, bad);
                su += u;
                ulo = fminf(ulo, u);
                uhi = fmaxf(uhi, u);
            } else {
                sp += r.z;
            }
        }
    };
    // The common case — periodic box, <= 4096 candidates, every displacement within one box length — without a branch
    // per candidate: the in-plane test only squares its components, so magnitudes do (|dx - copysign(L, dx)| =
    // |L - |dx|| bit for bit), membership becomes a select mask, and whatever would need the literal image loops is
    // only FLAGGED; if any lane raised the flag the sums are thrown away and the general code below runs instead.
    if (!done && flat && pbc) {
        const float halfa = La / 2.0f, halfb = Lb / 2.0f, halfn = Ln / 2.0f;
        bool redo = false;
        for (uint32_t it0 = 0; it0 < n_it; it0 += 4u) {
            LocalRec r[4];
            bool v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t iter = min(it0 + u, 63u);
                const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)it_base, (int)iter) + lane;
                v[u] = q < (uint32_t)__builtin_amdgcn_readlane((int)it_end, (int)iter);   // lanes >= n_it hold 0: never
                const uint32_t qc = v[u] ? q : 0u;
                r[u] = rec[qc];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const float ea = __builtin_fabsf(r[u].x - ha_pos), eb = __builtin_fabsf(r[u].y - hb_pos);
                const float ta = La - ea, tb = Lb - eb;
                const float ma = ea > halfa ? ta : ea, mb = eb > halfb ? tb : eb;
                const bool in = v[u] & (ma * ma + mb * mb < thr);            // `&`, `|`: no short-circuit branches
                redo |= v[u] & ((ta < 0.0f) | (tb < 0.0f));
                if (in) {      // one branch per candidate: a wave whose 64 candidates all lie outside skips the rest
                    const float dz = r[u].z - hn_pos;
                    const float uz = __builtin_fabsf(dz) > halfn ? dz - __builtin_copysignf(Ln, dz) : dz;
                    redo |= __builtin_fabsf(uz) > halfn;
                    cnt += 1u;
                    nf |= (r[u].z - r[u].z == 0.0f) ? 0u : 1u;
                    float sn, cs;
                    local_trig(r[u].z, inv_Ln, &sn, &cs);
                    sc += cs;
                    ss += sn;
                    su += uz;
                    ulo = __builtin_fminf(ulo, uz);
                    uhi = __builtin_fmaxf(uhi, uz);
                }
            }
        }
        done = !__any(redo);
        if (!done) { sc = ss = sp = su = 0.0f; ulo = 3.0e38f; uhi = -3.0e38f; cnt = 0; nf = 0; }
    }
    if (done) {
    } else if (flat) {
        for (uint32_t it0 = 0; it0 < n_it; it0 += 4u) {
            LocalRec r[4];
            bool v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t iter = min(it0 + u, 63u);
                const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)it_base, (int)iter) + lane;
                v[u] = q < (uint32_t)__builtin_amdgcn_readlane((int)it_end, (int)iter);   // lanes >= n_it hold 0: never
                const uint32_t qc = v[u] ? q : 0u;
                r[u] = rec[qc];
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++)
                if (v[u]) take(r[u]);
        }
    } else {
        for (uint32_t ia = 0; ia < n_rows; ia++) {
            const uint32_t row = ((a0 + ia) % nca) * ncs;
            for (uint32_t part = 0; part < 2u; part++) {
                const uint32_t q0 = part == 0 ? cstart[row + b0] : cstart[row];
                const uint32_t q1 = part == 0 ? cstart[row + b1] : cstart[row + b2];
                for (uint32_t q = q0 + lane; q < q1; q += 64u) take(rec[q]);
            }
        }
    }
    const double tcnt = (double)wave_total_u32(cnt);
    if (tcnt == 0.0 || __any(nf != 0u)) {
        if (lane == 0) raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, f, kStageTypes, a.mol_slot0 ? a.mol_slot0[m] : 0u, 0, m);
        return;
    }
    float center;
    if (!pbc) {
        center = (float)(wave_total((double)sp) / tcnt);
    } else {
        const float tc = wave_total_f32(sc), ts = wave_total_f32(ss);        // the estimate only anchors the image choice
        const float est = (atan2f(-ts, -tc) + 3.1415927f) / (6.2831855f / Ln);
        ulo = wave_min_f32(ulo);
        uhi = wave_max_f32(uhi);
        const float shift = gm_min_image(hn_pos - est, Ln, bad), half = Ln / 2.0f, margin = 1e-4f * Ln;
        const bool one_pass = ulo + shift > -half + margin && uhi + shift < half - margin;   // wave-uniform
        // pass 2 (only for a membrane thicker than half the box): refine with the mean minimum-image displacement
        // of the members from the estimate
        float ref = 0.0f;
        if (!one_pass) {       // rare: membership is simply tested again, run by run
            for (uint32_t ia = 0; ia < n_rows; ia++) {
                const uint32_t row = ((a0 + ia) % nca) * ncs;
                for (uint32_t part = 0; part < 2u; part++) {
                    const uint32_t q0 = part == 0 ? cstart[row + b0] : cstart[row];
                    const uint32_t q1 = part == 0 ? cstart[row + b1] : cstart[row + b2];
                    for (uint32_t q = q0 + lane; q < q1; q += 64u) {
                        const LocalRec r = rec[q];
                        if (inside(r.x, r.y)) ref += gm_min_image(r.z - est, Ln, bad);
                    }
                }
            }
        }
        if (one_pass) center = gm_wrap((est + shift) + (float)(wave_total((double)su) / tcnt), Ln, bad);
        else center = gm_wrap(est + (float)(wave_total((double)ref) / tcnt), Ln, bad);
    }
    if (lane == 0) {
        if (center != center) {
            raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, f, kStageTypes, a.mol_slot0 ? a.mol_slot0[m] : 0u, 0, m);
            return;
        }
        float d = hn_pos - center;
        if (pbc) d = gm_min_image(d, Ln, bad);
        a.aflags[(size_t)(a.row0 + s) * a.n_mol_total + m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
        if ((int)s == a.write_dist_frame && a.adist) a.adist[m] = d;
    }
    if (bad) raise_box_range(a.err, f);
}

// block = 256 threads = 4 waves = 4 heads; grid = (ceil(n_mol / 4), n_slab): the general passes for every head
// (no cell sums: NoPBC, or GORDER_HIP_LOCAL_ATOMS_ONLY)
__global__ __launch_bounds__(256) void k_local_flags(LocalArgs a) {
    const uint32_t m = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (m >= a.n_mol_total) return;
    local_flags_head(a, blockIdx.y, m);
}

// ---- a head per ROW of 16 lanes; the cells wholly inside the cylinder as spans of row prefix sums ------------------------
// The cells around a head (radius / 7 wide, 15 x 15 of them) are classified against the cylinder with the head's true
// position inside its cell, a ROW OF CELLS PER LANE: along a row the cells whose farthest corner is closer than the
// radius form one span of consecutive cells — taken as a block, the difference of two entries of k_local_rowprefix's
// running sums —, the cells the circle crosses form at most one run on each side of that span (one run altogether
// where the row has no inner cell), and the rest is outside.  Only the atoms of those ring runs are tested one by
// one: runs are contiguous in the record array (the halo keeps them so across the periodic boundary), cut into pieces
// of <= 8 records, two pieces per row and iteration.  Membership itself stays the exact distance test; the margins
// (4e-4 of r^2 either way) only decide who is tested.  Every periodic image of a neighbourhood cell other than the direct
// one lies beyond the radius (at least k cells of >= 1.0001 r / k away, local_axis), so the direct displacement is the
// minimum image here — unless a coordinate sits outside the box, which the flag `redo` catches.
// The block sum of u = MI(z - z_head) over an inner span is  sum z - n z_head  when no atom of the FRAME needs a shift
// relative to this head (finfo: the membrane's extrema of the normal coordinate; a membrane thicker than half the box,
// or a frame with a non-finite coordinate, sends the head to the general passes).
// FOUR heads per wave, one per DPP row: everything that is per head — the head's cell, the sums over the lanes (row
// shifts only), the centre, atan2f — is paid once per four heads.  A head the rows cannot decide is put on the list of
// k_local_flags_todo.
// block = 256 threads = 4 waves = 16 heads; grid = ceil(n_mol / 16) * (n_slab rounded up to 8), one dimension (see the
// XCD mapping below).  Periodic boxes, halo layout.
// atan2 to ~1e-5 rad (minimax polynomial of atan on [0, 1], hardware reciprocal); (0, 0) -> 0
__device__ __forceinline__ float local_atan2_fast(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    const float t = hi > 0.0f ? lo * __builtin_amdgcn_rcpf(hi) : 0.0f, s = t * t;
    float r = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0.0208351f, s, -0.0851330f), s, 0.1801410f), s, -0.3302995f), s, 0.9998660f) * t;
    r = ay > ax ? 1.5707964f - r : r;
    r = x < 0.0f ? 3.1415927f - r : r;
    return y < 0.0f ? -r : r;
}
#ifndef GORDER_ROW_FLIGHT
#define GORDER_ROW_FLIGHT 4
#endif
constexpr uint32_t kRowFlight = GORDER_ROW_FLIGHT;            // iterations of the ring loop whose records are fetched together
constexpr uint32_t kRowRing = 96;              // ring pieces (<= 8 records each) a head may list (typically ~45)
typedef uint2 LocalRingLists[kRowRing + 2u * kRowFlight];          // (the ring loop reads past a list's end, and masks)
// the 16 heads of workgroup `bx` of slab frame `s`
__device__ __forceinline__ void local_rows_group(const LocalArgs &a, uint32_t s, uint32_t bx, LocalRingLists *l_ring) {
    const uint32_t lane = threadIdx.x & 63u, row = lane >> 4, sub = lane & 15u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t m_raw = (bx * 4u + wave) * 4u + row;
    if ((bx * 4u + wave) * 4u >= a.n_mol_total) return;                   // the whole wave is past the last head
    const bool head_ok = m_raw < a.n_mol_total;
    const uint32_t m = head_ok ? m_raw : a.n_mol_total - 1u;              // idle rows shadow the last head, write nothing
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    const uint4 g = a.grid[s];
    const uint4 fk = reinterpret_cast<const uint4 *>(a.finfo)[s];
    const float z_min = local_key_float(fk.x), z_max = local_key_float(fk.y);
    const uint32_t nca = g.x, ncb = g.y, ka = g.z, kb = g.w, ncs = ncb + 2u * kb;
    const int da = (int)((a.dim + 1u) % 3u), db = (int)((a.dim + 2u) % 3u), dn = (int)a.dim;
    const float La = box[da], Lb = box[db], Ln = box[dn];
    const float halfn = Ln / 2.0f;
    const float thr = a.radius_thr;
    const uint32_t n_rows = 2u * ka + 1u;
    const LocalRec *rec = reinterpret_cast<const LocalRec *>(a.trig) + (size_t)s * a.rec_stride;
    const LocalRowPre *pre = a.rowpre + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u));
    const LocalEdge *edge = a.edge + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u));
    const uint32_t *cstart = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    // (uniform) grids the rows do not handle, frames with a non-finite coordinate
    const bool fail = !(ka >= 1u && kb >= 1u && n_rows <= 16u && fk.z == 0u && fk.x <= fk.y);

    // ---- the head of this row
    const float *hp = a.xyz + ((size_t)f * a.n_atoms + a.heads[m]) * 3u;
    const float ha_pos = hp[da], hb_pos = hp[db], hn_pos = hp[dn];
    int bad = 0;
    // gm_wrap's first step as selects (a head more than a box length outside leaves [0, L]: the general passes)
    const float wa = ha_pos < 0.0f ? ha_pos + La : (ha_pos > La ? ha_pos - La : ha_pos);
    const float wb = hb_pos < 0.0f ? hb_pos + Lb : (hb_pos > Lb ? hb_pos - Lb : hb_pos);
    // cell edges and the head's cell by hardware reciprocals (1 ulp): the head may land in the cell next to the one
    // k_local_build's arithmetic would give when it sits within an ulp of an edge — `fa`, `fb` below then lie that much
    // outside [0, edge], which the span bounds allow for (1e-3 of a cell) and the test below accepts (1e-4)
    const float ca = La * __builtin_amdgcn_rcpf((float)nca), cb = Lb * __builtin_amdgcn_rcpf((float)ncb);
    const uint32_t ha = (uint32_t)fminf(fmaxf(floorf(wa * __builtin_amdgcn_rcpf(ca)), 0.0f), (float)(nca - 1u));
    const uint32_t hb = (uint32_t)fminf(fmaxf(floorf(wb * __builtin_amdgcn_rcpf(cb)), 0.0f), (float)(ncb - 1u));
    const float fa = wa - (float)ha * ca, fb = wb - (float)hb * cb;        // the head inside its cell
    uint32_t a0 = ha + nca - ka, b0 = hb + ncb - kb;
    a0 -= a0 >= nca ? nca : 0u;
    b0 -= b0 >= ncb ? ncb : 0u;
    // a head outside the box by more than rounding, or not where its cell says: the general passes decide
    const float ulo_g = z_min - hn_pos, uhi_g = z_max - hn_pos;
    bool redo = !(fa >= -1e-4f * ca && fa <= ca * 1.0001f && fb >= -1e-4f * cb && fb <= cb * 1.0001f) ||
                !(wa >= 0.0f && wa <= La && wb >= 0.0f && wb <= Lb);

    // ---- this lane's row of cells: the inner span and the ring runs on either side
    float sc = 0.0f, ss = 0.0f, su = 0.0f;            // (sc, ss: over the cells wholly inside only, see the end)
    double inner_z = 0.0;
    uint32_t cnt = 0, n_inner = 0, n_ring = 0;
    // for the bound that may decide the head without its ring (below): this row's candidates — every record of the cells the
    // circle touches —, their number, the sum of their normal coordinates and of the squares, relative to the middle of the box
    const float z_mid = 0.5f * Ln;
    // (a frame whose atoms span more than three quarters of the box along the normal cannot pass condition (i) below with a
    // ring of the usual size: its heads do not try — a shortcut about speed, not about correctness)
    const bool try_prune = a.prune == 1 && z_max - z_min < 0.75f * Ln;       // (uniform; prune == 2: see k_local_decide)
    uint32_t c_n = 0, r_n = 0, j_lo = 0, j_hi = 0, row_e0 = 0;
    float c_z = 0.0f, r_z = 0.0f, r_q = 0.0f, c_c = 0.0f, c_s = 0.0f;
    bool span_ok = true;
    uint2 *ring = l_ring[wave * 4u + row];
    uint32_t run_q0[2] = {0u, 0u}, run_q1[2] = {0u, 0u};
    if (!fail && sub < n_rows) {
        const float r_in = thr * (1.0f - 4e-4f), r_out = thr * (1.0f + 4e-4f);
        const float a_lo = ((float)sub - (float)ka) * ca - fa, a_hi = a_lo + ca;      // the row's strip relative to the head
        const float a_far = fmaxf(fabsf(a_lo), fabsf(a_hi));
        const float a_near = (a_lo <= 0.0f && a_hi >= 0.0f) ? 0.0f : fminf(fabsf(a_lo), fabsf(a_hi));
        const float w_out2 = r_out - a_near * a_near, w_in2 = r_in - a_far * a_far;
        if (w_out2 > 0.0f) {
            // cell j of the row covers b in [(j - kb) cb - fb, (j - kb + 1) cb - fb]; with t = b / cb + kb (cell units
            // from the first cell's lower edge) cell j is [j, j + 1]
            // (1-ulp hardware square root and reciprocal: the bounds below carry 1e-3 of a cell of slack, the margins of
            // r_in / r_out 4e-4 of r^2)
            const float inv_cb = __builtin_amdgcn_rcpf(cb), t0 = fb * inv_cb + (float)kb;
            const float w_out = __builtin_amdgcn_sqrtf(w_out2) * inv_cb;
            // cells the circle may touch: j + 1 > t0 - w_out and j < t0 + w_out (one cell more on either side when the
            // bound is within rounding of a cell edge)
            const float jo_lo_f = floorf(t0 - w_out - 1e-3f), jo_hi_f = floorf(t0 + w_out + 1e-3f);
            const uint32_t jo_lo = (uint32_t)fminf(fmaxf(jo_lo_f, 0.0f), (float)(2u * kb));
            const uint32_t jo_hi = (uint32_t)fminf(fmaxf(jo_hi_f, 0.0f), (float)(2u * kb));
            // cells wholly inside: j >= t0 - w_in and j + 1 <= t0 + w_in (one cell less on either side near an edge)
            uint32_t ji_lo = 1u, ji_hi = 0u;                                          // empty
            if (w_in2 > 0.0f) {
                const float w_in = __builtin_amdgcn_sqrtf(w_in2) * inv_cb;
                const float lo_f = ceilf(t0 - w_in + 1e-3f), hi_f = floorf(t0 + w_in - 1e-3f) - 1.0f;
                if (hi_f >= lo_f) {
                    ji_lo = (uint32_t)fmaxf(lo_f, (float)jo_lo);
                    ji_hi = (uint32_t)fminf(hi_f, (float)jo_hi);
                }
            }
            uint32_t ra = a0 + sub;
            ra -= ra >= nca ? nca : 0u;
            const bool inner = ji_lo <= ji_hi;
            // the four cell edges of the row in ONE round trip: the outer span's two, the inner span's two (a row without an
            // inner cell takes the outer span's end for both: empty differences); an entry holds the cell's first record too
            const uint32_t j_d = jo_hi + 1u;
            j_lo = inner ? ji_lo : j_d;
            j_hi = inner ? ji_hi + 1u : j_d;
            row_e0 = ra * (ncs + 1u) + b0;
            uint32_t qa0, qb1, qa1, qb0;
            if (try_prune) {
                const LocalEdge *row_edge = edge + row_e0;
                const LocalEdge e_a = row_edge[jo_lo], e_d = row_edge[j_d], e_lo = row_edge[j_lo], e_hi = row_edge[j_hi];
                qa0 = e_a.q; qb1 = e_d.q; qa1 = e_lo.q; qb0 = e_hi.q;
                c_n = qb1 - qa0;
                c_z = e_d.zm - e_a.zm;
                r_n = c_n - (qb0 - qa1);
                r_z = c_z - (e_hi.zm - e_lo.zm);
                r_q = (e_d.sq - e_a.sq) - (e_hi.sq - e_lo.sq);
                local_edge_trig_diff(e_d.cs, e_a.cs, c_c, c_s);
                span_ok = c_n < kEdgeSpanMax;
            } else {                            // the exact path for certain: the cell list's own starts (4 bytes an edge) and its sums
                const uint32_t c0 = ra * ncs + b0;
                qa0 = cstart[c0 + jo_lo]; qb1 = cstart[c0 + j_d]; qa1 = cstart[c0 + j_lo]; qb0 = cstart[c0 + j_hi];
                if (inner) {
                    const LocalRowPre p_lo = pre[row_e0 + j_lo], p_hi = pre[row_e0 + j_hi];
                    sc += p_hi.sc - p_lo.sc;
                    ss += p_hi.ss - p_lo.ss;
                    inner_z = p_hi.sz - p_lo.sz;
                }
            }
            n_inner = qb0 - qa1;
            cnt += n_inner;
            run_q0[0] = qa0; run_q1[0] = qa1;
            run_q0[1] = qb0; run_q1[1] = qb1;
        }
    }
    // ---- a head its ring cannot change.  The leaflet is the SIGN of  z_head - centre,  centre = the mean normal coordinate of
    // the members (of their periodic images next to the circular-mean estimate).  Write u = z - z_head (plain differences) and
    // S = sum over the members of u: the sign wanted is minus the sign of S wherever (i) the reference's image choice moves
    // every member by the SAME number of box lengths and (ii) |S / n| < L / 2.
    // The members are the candidates (every record of the cells the circle touches: n_c of them, T = sum of their u, from the
    // prefix sums) without the non-members, which are SOME of the ring candidates (N of them, A = sum of u, B = sum of u^2,
    // prefix sums again).  For any subset K of the ring  sum_K u = (A + sum_j e_j u_j) / 2  with signs e_j = +-1, and
    // |sum_j e_j u_j| <= sqrt(N B) (Cauchy-Schwarz), so
    //     S  lies in  [T - A / 2 - sqrt(N B) / 2,  T - A / 2 + sqrt(N B) / 2] .
    // If that interval keeps clear of zero — by more than the rounding of the reference's f32 mean and of the f32 arithmetic
    // here can move it (`slack`) — the head's side is known without looking at a single ring atom.  That is the case for
    // every head whose local centre is not within a few hundredths of a nanometre of the head itself (measured: all heads of
    // the reference's own coarse-grained membrane and of the synthetic ones, flat or undulating by 2 nm;
    // tools/local_prune_probe.py).
    // (i): the circular mean of the CANDIDATES stands in for the reference's estimate (of the members) the way the inner
    // cells' does further down: the resultants differ by at most N unit vectors, the directions by asin(N / |R_c|), a length
    // of asin(N / |R_c|) L / (2 pi); every coordinate of the frame must keep that distance, and a margin, from the far side of
    // the box as seen from the stand-in.  (This is the condition that fails first: a membrane that undulates by a nanometre
    // in a 10-nm box leaves too little water for it, and its heads take the exact path, as before.)  (ii): (|T - A / 2| + sqrt(N B) / 2) / n_inner < L / 2.
    // A wave whose four heads are all decided skips the lists, the ring loop and the centre; any other wave does everything
    // as before (and finds the same sides).
    if (try_prune) {
        c_n = row_add_u32<0x111>(c_n); r_n = row_add_u32<0x111>(r_n); c_z = row_add<0x111>(c_z); r_z = row_add<0x111>(r_z); r_q = row_add<0x111>(r_q);
        c_n = row_add_u32<0x112>(c_n); r_n = row_add_u32<0x112>(r_n); c_z = row_add<0x112>(c_z); r_z = row_add<0x112>(r_z); r_q = row_add<0x112>(r_q);
        c_n = row_add_u32<0x114>(c_n); r_n = row_add_u32<0x114>(r_n); c_z = row_add<0x114>(c_z); r_z = row_add<0x114>(r_z); r_q = row_add<0x114>(r_q);
        c_n = row_add_u32<0x118>(c_n); r_n = row_add_u32<0x118>(r_n); c_z = row_add<0x118>(c_z); r_z = row_add<0x118>(r_z); r_q = row_add<0x118>(r_q);
        c_c = row_add<0x111>(c_c); c_s = row_add<0x111>(c_s);
        c_c = row_add<0x112>(c_c); c_s = row_add<0x112>(c_s);
        c_c = row_add<0x114>(c_c); c_s = row_add<0x114>(c_s);
        c_c = row_add<0x118>(c_c); c_s = row_add<0x118>(c_s);
        const float hm = hn_pos - z_mid, fn = (float)c_n, fr = (float)r_n, fi = (float)(c_n - r_n);
        const float T = c_z - fn * hm, A = r_z - fr * hm;
        const float B = __builtin_fmaxf((r_q - 2.0f * hm * r_z) + fr * hm * hm, 0.0f) * 1.02f + 1e-3f * fr;
        const float mid = T - 0.5f * A, rad = 0.5005f * __builtin_amdgcn_sqrtf(fr * B);
        const float slack = 0.1f + fn * (1e-3f + 1e-6f * fn * Ln);
        // (i)
        const float r_c = __builtin_amdgcn_sqrtf(c_c * c_c + c_s * c_s);
        const float est_c = (local_atan2_fast(-c_s, -c_c) + 3.1415927f) * (Ln * 0.15915494f);
        const float shift_c = gm_min_image(hn_pos - est_c, Ln, bad);
        const float xr = (fr + 1.0f) * __builtin_amdgcn_rcpf(r_c);          // asin(x) <= x + (pi / 2 - 1) x^3 on [0, 1]
        const float emargin = 1e-4f * Ln + (xr + 0.5708f * xr * xr * xr) * (0.15916f * Ln);
        const bool same_image = fr + 1.0f < r_c && ulo_g + shift_c > -halfn + emargin && uhi_g + shift_c < halfn - emargin;
        // (ii)
        const bool head_near = __builtin_fabsf(mid) + rad < (halfn - 1e-3f * Ln) * fi;
        const uint64_t wide = __ballot(!span_ok);        // a span too long for the 16-bit cos / sin differences: its head is not decided here
        const bool spans_ok = ((wide >> (16u * row)) & 0xffffull) == 0ull;
        const bool decided = !fail && !redo && spans_ok && c_n > r_n && same_image && head_near && (int)s != a.write_dist_frame &&
                             __builtin_fabsf(mid) > rad + slack;             // (NaN anywhere: not decided)
        const uint64_t dm = __ballot(sub != 15u || decided || !head_ok);
        if (dm == ~0ull) {
            if (sub == 15u && head_ok)       // S > 0: the centre lies above the head, d = z_head - centre < 0
                a.aflags[(size_t)(a.row0 + s) * a.n_mol_total + m] = (uint8_t)((mid > 0.0f ? 1 : 0) ^ (a.flip ? 1 : 0));
            if (bad) raise_box_range(a.err, f);
            return;
        }
    }
    // ---- the exact path from here: the inner span's sums for the centre
    if (try_prune && !fail && sub < n_rows && j_lo != j_hi) {
        const LocalRowPre p_lo = pre[row_e0 + j_lo], p_hi = pre[row_e0 + j_hi];
        sc += p_hi.sc - p_lo.sc;
        ss += p_hi.ss - p_lo.ss;
        inner_z = p_hi.sz - p_lo.sz;
    }
    // ---- the ring runs go to the row's list in pieces of <= 8 records: every lane's share of the list starts where
    // the lanes before it in the row end (exclusive scan of the piece counts by row shifts)
    {
        const uint32_t n_a = run_q1[0] - run_q0[0], n_b = run_q1[1] - run_q0[1];
        const uint32_t p_a = (n_a + 7u) >> 3, p_b = (n_b + 7u) >> 3, mine = p_a + p_b;
        uint32_t incl = mine;
        incl = row_add_u32<0x111>(incl); incl = row_add_u32<0x112>(incl); incl = row_add_u32<0x114>(incl); incl = row_add_u32<0x118>(incl);
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
        const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 47), t3 = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        n_ring = row == 0u ? t0 : (row == 1u ? t1 : (row == 2u ? t2 : t3));
        // (a piece = the BYTE offsets of its first record and of the end of its run: the loop below adds a lane's share)
        constexpr uint32_t kRec = (uint32_t)sizeof(LocalRec);
        const uint32_t pos = incl - mine;
        const uint32_t a_from = run_q0[0] * kRec, a_to = run_q1[0] * kRec, b_from = run_q0[1] * kRec, b_to = run_q1[1] * kRec;
        for (uint32_t k = 0; k < mine; k++) {          // (one loop over both runs: the wave makes the longest lane's trips)
            const bool second = k >= p_a;
            const uint32_t from = (second ? b_from - p_a * (8u * kRec) : a_from) + k * (8u * kRec), to = second ? b_to : a_to;
            if (pos + k < kRowRing) ring[pos + k] = make_uint2(from, min(to, from + 8u * kRec));
        }
    }
    redo |= n_ring > kRowRing;                                  // the list ran over: the general passes
    // ---- the ring: two pieces per row and iteration (lanes 0-7 and 8-15), one record per lane, the records of
    // kRowFlight iterations fetched together.  The loop runs to the longest of the wave's four lists; the shorter ones are
    // filled up with empty pieces, so that a lane's record is in its piece exactly when its offset is below the piece's
    // end — the only test —, and the records come by buffer loads: 32-bit offsets from one scalar descriptor of this
    // frame's records, no address arithmetic, and an offset past the frame's last record (the lanes past the end of the
    // last piece) reads zeros instead of the neighbour's memory.
    uint32_t n_max = max(max((uint32_t)__builtin_amdgcn_readlane((int)n_ring, 0), (uint32_t)__builtin_amdgcn_readlane((int)n_ring, 16)),
                         max((uint32_t)__builtin_amdgcn_readlane((int)n_ring, 32), (uint32_t)__builtin_amdgcn_readlane((int)n_ring, 48)));
    n_max = min(n_max, kRowRing);
    const uint32_t n_mine = min(n_ring, kRowRing), half = sub >> 3, lane_off = (sub & 7u) * (uint32_t)sizeof(LocalRec);
    const uint32_t n_end = (n_max + 2u * kRowFlight - 1u) / (2u * kRowFlight) * (2u * kRowFlight);
    for (uint32_t e = n_mine + sub; e < n_end; e += 16u) ring[e] = make_uint2(0u, 0u);
    __builtin_amdgcn_wave_barrier();
    const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<LocalRec *>(rec), 0, a.rec_stride * (uint32_t)sizeof(LocalRec), 0x00020000);
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    typedef float f3 __attribute__((ext_vector_type(3)));
    float t_min = 0.0f;                 // the smallest of the L - |d| seen (negative: a displacement of more than a box)
    for (uint32_t t = 0; !fail && t < n_max; t += 2u * kRowFlight) {
        f3 r[kRowFlight];
        uint2 run[kRowFlight];
        bool v[kRowFlight];
#pragma unroll
        for (uint32_t u = 0; u < kRowFlight; u++) run[u] = ring[t + 2u * u + half];
#pragma unroll
        for (uint32_t u = 0; u < kRowFlight; u++) {
            const uint32_t q = run[u].x + lane_off;
            v[u] = q < run[u].y;
            r[u] = __builtin_bit_cast(f3, (u3)__builtin_amdgcn_raw_buffer_load_b96(rec_rsrc, (int)q, 0, 0));
        }
#pragma unroll
        for (uint32_t u = 0; u < kRowFlight; u++) {
            // |d - copysign(L, d)| = L - |d| for |d| > L / 2, and L - |d| < |d| exactly then (L / 2 is exact, the rounding of
            // the difference monotonic): the one-step minimum image's magnitude is the smaller of the two.
            // (Lanes without a record test whatever they fetched — a record of the frame, or zeros — and are masked; that
            // such a record may raise the more-than-a-box flag only sends a head to the general passes needlessly.)
            const float ea = __builtin_fabsf(r[u].x - ha_pos), eb = __builtin_fabsf(r[u].y - hb_pos);
            const float ta = La - ea, tb = Lb - eb;
            const float ma = __builtin_fminf(ea, ta), mb = __builtin_fminf(eb, tb);
            const bool in = v[u] & (ma * ma + mb * mb < thr);
            t_min = __builtin_fminf(t_min, __builtin_fminf(ta, tb));
            cnt += in ? 1u : 0u;
            su += in ? r[u].z : 0.0f;                   // (the coordinate as it is: see the end)
        }
    }
    redo |= t_min < 0.0f;
    // ---- per row: totals in lane 15 of the row (row shifts only), then the centre as in the general passes
    double tu = (double)su + inner_z;
    tu = row_add_f64<0x111>(tu); tu = row_add_f64<0x112>(tu); tu = row_add_f64<0x114>(tu); tu = row_add_f64<0x118>(tu);
    sc = row_add<0x111>(sc); ss = row_add<0x111>(ss); cnt = row_add_u32<0x111>(cnt); n_inner = row_add_u32<0x111>(n_inner);
    sc = row_add<0x112>(sc); ss = row_add<0x112>(ss); cnt = row_add_u32<0x112>(cnt); n_inner = row_add_u32<0x112>(n_inner);
    sc = row_add<0x114>(sc); ss = row_add<0x114>(ss); cnt = row_add_u32<0x114>(cnt); n_inner = row_add_u32<0x114>(n_inner);
    sc = row_add<0x118>(sc); ss = row_add<0x118>(ss); cnt = row_add_u32<0x118>(cnt); n_inner = row_add_u32<0x118>(n_inner);
    const uint64_t redo_mask = __ballot(redo);
    const bool row_redo = ((redo_mask >> (16u * row)) & 0xffffull) != 0ull;
    bool general = fail || row_redo;
    if (sub == 15u && head_ok && !general) {
        if (cnt == 0u) {
            raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, f, kStageTypes, a.mol_slot0 ? a.mol_slot0[m] : 0u, 0, m);
        } else {
            // The reference's estimate (the circular mean of ALL members) only picks, per member, the image next to it;
            // the centre is the plain mean of those images.  Here the circular mean of the members in the cells wholly
            // inside stands in for it — prefix sums, no sine and cosine per ring record —, with a bound on how far the
            // two can lie apart: the resultants differ by the ring members' unit vectors, |R_all - R_inner| <= e = the
            // number of ring members, so the directions differ by at most asin(e / |R_inner|) <= (pi / 2) e / |R_inner|,
            // a length of (e / |R_inner|) L / 4 along the normal.  If every coordinate of the frame (a superset of the
            // members) keeps that distance, and a margin, from the far side of the box as seen from the stand-in, each
            // member's image next to the reference's estimate is its own coordinate moved by the SAME number of box
            // lengths, and the centre is the wrapped plain mean.  (The +1 covers the rounding of the f32 prefix
            // differences, the 1e-5 rad arctangent and the hardware sine / cosine, generously.)
            const float r_inner = __builtin_amdgcn_sqrtf(sc * sc + ss * ss), e_ring = (float)(cnt - n_inner) + 1.0f;
            const float est = (local_atan2_fast(-ss, -sc) + 3.1415927f) * (Ln * 0.15915494f);
            const float shift = gm_min_image(hn_pos - est, Ln, bad);
            const float margin = 1e-4f * Ln + e_ring * __builtin_amdgcn_rcpf(r_inner) * (0.2501f * Ln);
            if (e_ring < r_inner && ulo_g + shift > -halfn + margin && uhi_g + shift < halfn - margin) {
                // (the quotient by a reciprocal and one Newton step: 2^-52 or so, a division's worth without its cost)
                const double dc = (double)cnt;
                double rc = __builtin_amdgcn_rcp(dc);
                rc = __builtin_fma(__builtin_fma(-dc, rc, 1.0), rc, rc);
                rc = __builtin_fma(__builtin_fma(-dc, rc, 1.0), rc, rc);
                const float center = gm_wrap((float)(tu * rc), Ln, bad);
                if (center != center) {
                    raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, f, kStageTypes, a.mol_slot0 ? a.mol_slot0[m] : 0u, 0, m);
                } else {
                    const float d = gm_min_image(hn_pos - center, Ln, bad);
                    a.aflags[(size_t)(a.row0 + s) * a.n_mol_total + m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
                    if ((int)s == a.write_dist_frame && a.adist) a.adist[m] = d;
                }
            } else {
                general = true;          // too thick a membrane for the bound: the general passes
            }
        }
    }
    if (bad) raise_box_range(a.err, f);
    // ---- heads the rows could not decide go on the list of k_local_flags_todo (calling the general passes from here
    // would cost this kernel their registers and a stack)
    if (general && head_ok && sub == 15u) {
        const uint32_t at = atomicAdd(&a.todo[0].x, 1u);
        a.todo[1u + at] = make_uint2(s, m);
    }
}
// Workgroups go to the 8 XCDs round-robin in launch order, and every XCD has an L2 of its own: the heads of ONE frame
// share that frame's records (a head reads ~330 ring records, neighbouring heads mostly the same ones), so all the
// workgroups of a frame are given to one XCD — launch index L -> XCD L mod 8 takes frames (L / 8) / groups * 8 + L mod 8 —
// and a frame's records come from HBM once instead of once per XCD (measured: 2.1 GB -> see DESIGN, K6).
// k_local_flags_rows: grid = rows_groups * (n_slab rounded up to 8), a workgroup per (frame, 16 heads).
// k_local_flags_rows_open, behind k_local_decide: grid = rows_groups * 8 * kRowsSlots, and the workgroups of XCD x walk the frames x, x + 8, ...
// that have a head left open (need[s] != 0) — slot j of kRowsSlots takes every kRowsSlots-th of them —: none, normally, and
// the launch costs what 1 536 workgroups cost that read one word each (a workgroup per frame and 16 heads that leaves at
// once: 40 us per 512 frames; 6 144 workgroups: 17 us); with every frame open a workgroup makes 64 trips, all of the chip's.
constexpr uint32_t kRowsSlots = 1;
__global__ __launch_bounds__(256) void k_local_flags_rows(LocalArgs a) {
    __shared__ LocalRingLists l_ring[16];
    const uint32_t n_groups = a.rows_groups, linear = blockIdx.x;
    const uint32_t xcd = linear & 7u, k = linear >> 3;
    const uint32_t s = (k / n_groups) * 8u + xcd, bx = k - (k / n_groups) * n_groups;
    if (s < a.n_slab) local_rows_group(a, s, bx, l_ring);                   // (the slab's frame count rounded up to 8)
}
__global__ __launch_bounds__(256) void k_local_flags_rows_open(LocalArgs a) {
    __shared__ LocalRingLists l_ring[16];
    const uint32_t n_groups = a.rows_groups, linear = blockIdx.x;
    const uint32_t xcd = linear & 7u, k = linear >> 3;
    if (a.need[a.n_slab] == 0u) return;                                     // no frame of the slab has a head left open (one scalar load)
    const uint32_t slot = k / n_groups, bx = k - slot * n_groups, lane = threadIdx.x & 63u;
    uint32_t seen = 0;
    for (uint32_t base = xcd; base < a.n_slab; base += 8u * 64u) {          // 64 of this XCD's frames at a time, a lane each
        const uint32_t sl = base + 8u * lane;
        uint64_t open = __ballot(sl < a.n_slab && a.need[sl] != 0u);
        while (open != 0ull) {                                              // (uniform)
            const uint32_t bit = (uint32_t)__builtin_ctzll(open);
            open &= open - 1ull;
            if (seen++ % kRowsSlots == slot) local_rows_group(a, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + 8u * bit)), bx, l_ring);
        }
    }
}

// The table of cell edges WITHOUT the cell list.  k_local_decide reads nothing else, and where it decides every head of a
// frame (the usual case) the sorted records, the cell starts and the f64 row sums of k_local_build / k_local_rowprefix —
// 60 % of the step — are made for nobody.  The edge entries are sums over cells, and sums do not need the atoms in order:
// a workgroup per frame adds every membrane atom into its cell's two 64-bit words in LDS (fixed point; one pass, two LDS
// atomics an atom, nothing kept per atom), then a wave per row of cells scans the row's cells — integers, so the order of the
// atomics does not show — and writes the entries, halo columns included.  Frames that keep a head open (need[s] != 0 after
// k_local_decide; the frame whose distances are exported is one) get their cell list and row sums from k_local_build and
// k_local_rowprefix as before — those kernels leave at once for the other frames — and k_local_rowprefix overwrites the
// frame's entries with its own.
//   word A = sum of rint((z - L/2 + 8) 2^15) << 32 | sum of rint((z - L/2)^2 2^13)          (|z - L/2| < 8 nm, or the frame is left open)
//   word B = count << 42 | sum of rint((cos + 1) 2^8) << 21 | sum of rint((sin + 1) 2^8)    (4 095 atoms a cell, or the frame is left open)
// block = 1024 threads; grid = n_slab; dynamic LDS = kSumsLds (frames with more than kSumsCells cells are left open).
constexpr uint32_t kSumsCells = 9000;
constexpr uint32_t kSumsLds = kSumsCells * 2u * (uint32_t)sizeof(unsigned long long);
// an atom more than a box length outside the box in the plane (rare): gm_wrap's loops, once for the whole kernel
struct LocalWrapped { float a, b; int bad; };
__device__ __attribute__((noinline)) LocalWrapped local_wrap_far(float xa, float xb, float La, float Lb) {
    LocalWrapped w;
    w.bad = 0;
    w.a = gm_wrap(xa, La, w.bad);
    w.b = gm_wrap(xb, Lb, w.bad);
    return w;
}
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v, uint32_t lane) {         // inclusive, over the 64 lanes
    v = row_add_u32<0x111>(v); v = row_add_u32<0x112>(v); v = row_add_u32<0x114>(v); v = row_add_u32<0x118>(v);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t r = lane >> 4;
    return v + (r > 0u ? t0 : 0u) + (r > 1u ? t1 : 0u) + (r > 2u ? t2 : 0u);
}
__device__ __forceinline__ double wave_scan_f64(double v, uint32_t lane) {             // (exact: the values are integers below 2^53)
    v = row_add_f64<0x111>(v); v = row_add_f64<0x112>(v); v = row_add_f64<0x114>(v); v = row_add_f64<0x118>(v);
    const double t0 = lane_value(v, 15), t1 = lane_value(v, 31), t2 = lane_value(v, 47);
    const uint32_t r = lane >> 4;
    return v + (r > 0u ? t0 : 0.0) + (r > 1u ? t1 : 0.0) + (r > 2u ? t2 : 0.0);
}
__global__ __launch_bounds__(1024) void k_local_sums(LocalArgs a) {
    extern __shared__ unsigned long long l_sums[];
    __shared__ float l_zlo[16], l_zhi[16];
    __shared__ uint32_t l_flag[16];
    const uint32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb, ka, kb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
    const uint32_t ncs = local_row_stride(a, ncb, kb), n_base = nca * ncb;
    const int dn = (int)a.dim;
    if (tid == 0u) a.grid[s] = make_uint4(nca, ncb, ka, kb);
    if (!(a.pbc && ka >= 1u && kb >= 1u && n_base <= kSumsCells)) {          // (uniform) not a grid for this kernel: the frame is left open
        if (tid == 0u) { local_finfo_init(a, s); a.need[s] = 1u; a.need[a.n_slab] = 1u; }
        return;
    }
    unsigned long long *l_a = l_sums, *l_b = l_sums + n_base;
    for (uint32_t k = tid; k < 2u * n_base; k += 1024u) l_sums[k] = 0ull;
    __syncthreads();
    const float *x = a.xyz + (size_t)f * a.n_atoms * 3u;
    const float La = box[da], Lb = box[db], Ln = box[dn];
    const float inv_a = (float)nca / La, inv_b = (float)ncb / Lb, inv_Ln = 1.0f / Ln, z_mid = 0.5f * Ln;
    int bad = 0;
    float zlo = 3.0e38f, zhi = -3.0e38f;
    uint32_t flag = 0;                                  // 1: a coordinate is not finite; 2: an atom the fixed point has no room for
    constexpr uint32_t U = 8;
    for (uint32_t i0 = tid; i0 < a.n_membrane; i0 += U * 1024u) {           // (uniform trip count up to the last trip's tail)
        uint32_t at[U];
        float pa[U], pb[U], pn[U];
#pragma unroll
        for (uint32_t u = 0; u < U; u++) { const uint32_t i = min(i0 + u * 1024u, a.n_membrane - 1u); at[u] = a.membrane ? a.membrane[i] : i; }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            pa[u] = x[3u * (size_t)at[u] + da]; pb[u] = x[3u * (size_t)at[u] + db]; pn[u] = x[3u * (size_t)at[u] + dn];
        }
#pragma unroll
        for (uint32_t u = 0; u < U; u++) {
            if (i0 + u * 1024u >= a.n_membrane) continue;
            // the atom's cell: k_local_build's arithmetic (cell_ab)
            float wa = pa[u] > La ? pa[u] - La : pa[u], wb = pb[u] > Lb ? pb[u] - Lb : pb[u];
            wa = wa < 0.0f ? wa + La : wa;
            wb = wb < 0.0f ? wb + Lb : wb;
            if (__builtin_expect(!(wa >= 0.0f && wa <= La && wb >= 0.0f && wb <= Lb), 0)) {
                const LocalWrapped w = local_wrap_far(pa[u], pb[u], La, Lb);           // (out of line: eight inlined copies of
                wa = w.a; wb = w.b; bad |= w.bad;                                        //  gm_wrap's loops were 70 % of this kernel's code)
            }
            const uint32_t ca = (uint32_t)fminf(fmaxf(floorf(wa * inv_a), 0.0f), (float)(nca - 1u));
            const uint32_t cb = (uint32_t)fminf(fmaxf(floorf(wb * inv_b), 0.0f), (float)(ncb - 1u));
            const float z = pn[u], zm = z - z_mid;
            float sn, cs;
            local_trig(z, inv_Ln, &sn, &cs);
            const bool finite = ((pa[u] - pa[u]) + (pb[u] - pb[u])) + (z - z) == 0.0f;
            const bool room = zm > -8.0f && zm < 8.0f;
            flag |= (finite ? 0u : 1u) | (room ? 0u : 2u);
            zlo = fminf(zlo, z);
            zhi = fmaxf(zhi, z);
            if (finite && room) {
                const unsigned long long wa64 = ((unsigned long long)(uint32_t)__builtin_rintf((zm + 8.0f) * 32768.0f) << 32) |
                                                (unsigned long long)(uint32_t)__builtin_rintf(zm * zm * 8192.0f);
                const unsigned long long wb64 = (1ull << 42) | ((unsigned long long)(uint32_t)__builtin_rintf((cs + 1.0f) * 256.0f) << 21) |
                                                (unsigned long long)(uint32_t)__builtin_rintf((sn + 1.0f) * 256.0f);
                atomicAdd(&l_a[ca * ncb + cb], wa64);
                atomicAdd(&l_b[ca * ncb + cb], wb64);
            }
        }
    }
    if (bad) raise_box_range(a.err, f);
    __syncthreads();
    // ---- a wave per row of cells: the entries = exclusive sums along the row, the first 2 kb cells again behind the last
    LocalEdge *out = a.edge + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u));
    for (uint32_t ra = wave; ra < nca; ra += 16u) {
        uint32_t c_n = 0, c_c = 0, c_s = 0;             // the sums of the chunks before (64 cells a chunk)
        double c_z = 0.0, c_q = 0.0;
        for (uint32_t j0 = 0; j0 <= ncs; j0 += 64u) {
            const uint32_t j = j0 + lane, jb = j < ncb ? j : j - ncb;
            unsigned long long va = 0ull, vb = 0ull;
            if (j < ncs) { va = l_a[ra * ncb + jb]; vb = l_b[ra * ncb + jb]; }
            const uint32_t n = (uint32_t)(vb >> 42), ci = (uint32_t)(vb >> 21) & 0x1fffffu, si = (uint32_t)vb & 0x1fffffu;
            const double zi = (double)(uint32_t)(va >> 32), qi = (double)(uint32_t)va;
            flag |= n > 4095u ? 2u : 0u;
            const uint32_t i_n = wave_scan_u32(n, lane), i_c = wave_scan_u32(ci, lane), i_s = wave_scan_u32(si, lane);
            const double i_z = wave_scan_f64(zi, lane), i_q = wave_scan_f64(qi, lane);
            if (j <= ncs) {
                const uint32_t e_n = c_n + (i_n - n);
                const double e_z = c_z + (i_z - zi), e_q = c_q + (i_q - qi);
                const float e_c = (float)(c_c + (i_c - ci)) * (1.0f / 256.0f) - (float)e_n;
                const float e_s = (float)(c_s + (i_s - si)) * (1.0f / 256.0f) - (float)e_n;
                LocalEdge e;
                e.q = e_n;              // (relative to the row's first cell: k_local_decide takes differences inside a row only)
                e.zm = (float)(e_z * (1.0 / 32768.0) - 8.0 * (double)e_n);
                e.sq = (float)(e_q * (1.0 / 8192.0));
                e.cs = local_edge_trig(e_c, e_s);
                out[(size_t)ra * (ncs + 1u) + j] = e;
            }
            c_n += (uint32_t)__builtin_amdgcn_readlane((int)i_n, 63);
            c_c += (uint32_t)__builtin_amdgcn_readlane((int)i_c, 63);
            c_s += (uint32_t)__builtin_amdgcn_readlane((int)i_s, 63);
            c_z += lane_value(i_z, 63);
            c_q += lane_value(i_q, 63);
        }
    }
    // ---- the frame's record: extrema of the normal coordinate, and whether the frame is for this kernel at all
    for (int off = 32; off >= 1; off >>= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, off, 64));
        zhi = fmaxf(zhi, __shfl_xor(zhi, off, 64));
        flag |= (uint32_t)__shfl_xor((int)flag, off, 64);
    }
    if (lane == 0u) { l_zlo[wave] = zlo; l_zhi[wave] = zhi; l_flag[wave] = flag; }
    __syncthreads();
    if (tid == 0u) {
        for (uint32_t w = 1; w < 16u; w++) { zlo = fminf(zlo, l_zlo[w]); zhi = fmaxf(zhi, l_zhi[w]); flag |= l_flag[w]; }
        reinterpret_cast<uint4 *>(a.finfo)[s] = zlo <= zhi ? make_uint4(local_float_key(zlo), local_float_key(zhi), flag & 1u, 2u)
                                                            : make_uint4(0xffffffffu, 0u, flag & 1u, 2u);
        if (flag) { a.need[s] = 1u; a.need[a.n_slab] = 1u; }          // (k_local_build makes the frame's record again, and everything else)
    }
}

// The bound of k_local_flags_rows ("a head its ring cannot change") on its own, a LANE per head, the frame's table of cell
// edges in the LDS.  In the rows kernel a head is the business of 16 lanes — one per row of cells, which is what its ring
// loop wants —, and everything that is per head (its cell, the centre's estimate, the decision: two thirds of the ~390
// instructions a wave spends on four heads) is done 16 times over: 152 M instructions per 512 frames, all the SIMDs' cycles
// (PMC).  A lane that walks the 15 rows of its own head spends ~2 000 instructions on 64 heads instead of 16 x 390 — and
// then waits for the loads: a head's 60 entries are 60 addresses of their own, a wave's load instruction 64 cache lines, and
// the texture path looks up one line a cycle (1.57 M such instructions per 512 frames = 164 us whichever way the lanes are
// dealt: the lane-per-head kernel took 192 us reading the table from memory, the rows kernel 238).  The table of ONE frame,
// though, is 84 x 99 entries of 16 bytes for the 3 072-lipid membrane — 133 KB, and a CU has 160 KB of LDS: a workgroup per
// frame copies it in once, coalesced, and the heads' lookups are LDS reads.
// A frame this kernel decides completely costs k_local_flags_rows_open nothing; a frame with a head left open
// (need[s] != 0) is done there as before, every head of it (the sides agree: both are the reference's).  A frame whose
// table does not fit is read from memory.
// block = 1024 threads, a lane per head, as many trips as the frame has heads; grid = n_slab; dynamic LDS = kDecideLds.
constexpr uint32_t kDecideEntries = 10000;                                  // table entries the LDS copy has room for
constexpr uint32_t kDecideLds = kDecideEntries * (uint32_t)sizeof(LocalEdge);
struct DecideFrame {
    uint32_t nca, ncb, ka, kb, ncs, n_rows;
    int da, db, dn;
    float La, Lb, Ln, z_min, z_max;
};
// one head; `edge` = the frame's table (LDS or memory).  true: decided, `upper_side` = its flag before the flip
template <typename EdgePtr>
__device__ __forceinline__ bool local_decide_head(const LocalArgs &a, const DecideFrame &F, uint32_t f, uint32_t m, EdgePtr edge,
                                                  bool &centre_above, int &bad) {
    const uint32_t nca = F.nca, ncb = F.ncb, ka = F.ka, kb = F.kb, ncs = F.ncs, n_rows = F.n_rows;
    const float La = F.La, Lb = F.Lb, Ln = F.Ln;
    const float halfn = Ln / 2.0f, z_mid = 0.5f * Ln;
    const float thr = a.radius_thr;
    // ---- the head (as in k_local_flags_rows)
    const float *hp = a.xyz + ((size_t)f * a.n_atoms + a.heads[m]) * 3u;
    const float ha_pos = hp[F.da], hb_pos = hp[F.db], hn_pos = hp[F.dn];
    const float wa = ha_pos < 0.0f ? ha_pos + La : (ha_pos > La ? ha_pos - La : ha_pos);
    const float wb = hb_pos < 0.0f ? hb_pos + Lb : (hb_pos > Lb ? hb_pos - Lb : hb_pos);
    const float ca = La * __builtin_amdgcn_rcpf((float)nca), cb = Lb * __builtin_amdgcn_rcpf((float)ncb);
    const uint32_t ha = (uint32_t)fminf(fmaxf(floorf(wa * __builtin_amdgcn_rcpf(ca)), 0.0f), (float)(nca - 1u));
    const uint32_t hb = (uint32_t)fminf(fmaxf(floorf(wb * __builtin_amdgcn_rcpf(cb)), 0.0f), (float)(ncb - 1u));
    const float fa = wa - (float)ha * ca, fb = wb - (float)hb * cb;
    uint32_t a0 = ha + nca - ka, b0 = hb + ncb - kb;
    a0 -= a0 >= nca ? nca : 0u;
    b0 -= b0 >= ncb ? ncb : 0u;
    const float ulo_g = F.z_min - hn_pos, uhi_g = F.z_max - hn_pos;
    const bool redo = !(fa >= -1e-4f * ca && fa <= ca * 1.0001f && fb >= -1e-4f * cb && fb <= cb * 1.0001f) ||
                      !(wa >= 0.0f && wa <= La && wb >= 0.0f && wb <= Lb);
    // ---- the rows of cells around the head, four at a time
    const float r_in = thr * (1.0f - 4e-4f), r_out = thr * (1.0f + 4e-4f);
    const float inv_cb = __builtin_amdgcn_rcpf(cb), t0 = fb * inv_cb + (float)kb, two_kb = (float)(2u * kb);
    uint32_t c_n = 0, r_n = 0;
    float c_z = 0.0f, r_z = 0.0f, r_q = 0.0f, c_c = 0.0f, c_s = 0.0f;
    bool spans_ok = true;
    for (uint32_t sub0 = 0; sub0 < n_rows; sub0 += 4u) {                   // (uniform)
        LocalEdge e_a[4], e_d[4], e_lo[4], e_hi[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t sub = sub0 + u;
            const float a_lo = ((float)sub - (float)ka) * ca - fa, a_hi = a_lo + ca;      // the row's strip relative to the head
            const float a_far = fmaxf(fabsf(a_lo), fabsf(a_hi));
            const float a_near = (a_lo <= 0.0f && a_hi >= 0.0f) ? 0.0f : fminf(fabsf(a_lo), fabsf(a_hi));
            const float w_out2 = r_out - a_near * a_near, w_in2 = r_in - a_far * a_far;
            const bool touch = sub < n_rows && w_out2 > 0.0f;
            // (the spans as in k_local_flags_rows; a row the circle does not touch reads entry 0 four times: empty differences)
            const float w_out = __builtin_amdgcn_sqrtf(fmaxf(w_out2, 0.0f)) * inv_cb;
            const float jo_lo_f = floorf(t0 - w_out - 1e-3f), jo_hi_f = floorf(t0 + w_out + 1e-3f);
            const uint32_t jo_lo = (uint32_t)fminf(fmaxf(jo_lo_f, 0.0f), two_kb);
            const uint32_t jo_hi = (uint32_t)fminf(fmaxf(jo_hi_f, 0.0f), two_kb);
            const float w_in = __builtin_amdgcn_sqrtf(fmaxf(w_in2, 0.0f)) * inv_cb;
            const float lo_f = ceilf(t0 - w_in + 1e-3f), hi_f = floorf(t0 + w_in - 1e-3f) - 1.0f;
            const bool inner = w_in2 > 0.0f && hi_f >= lo_f;
            const uint32_t ji_lo = (uint32_t)fmaxf(lo_f, (float)jo_lo), ji_hi = (uint32_t)fmaxf(fminf(hi_f, (float)jo_hi), 0.0f);
            const bool has_inner = inner && ji_lo <= ji_hi;
            const uint32_t j_d = jo_hi + 1u;
            const uint32_t j_lo = has_inner ? ji_lo : j_d, j_hi = has_inner ? ji_hi + 1u : j_d;
            uint32_t ra = a0 + (sub < n_rows ? sub : 0u);
            ra -= ra >= nca ? nca : 0u;
            EdgePtr row_edge = edge + (ra * (ncs + 1u) + b0);
            e_a[u] = row_edge[touch ? jo_lo : 0u];
            e_d[u] = row_edge[touch ? j_d : 0u];
            e_lo[u] = row_edge[touch ? j_lo : 0u];
            e_hi[u] = row_edge[touch ? j_hi : 0u];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) {
            const uint32_t n_c = e_d[u].q - e_a[u].q;
            const float z_c = e_d[u].zm - e_a[u].zm;
            c_n += n_c;
            c_z += z_c;
            r_n += n_c - (e_hi[u].q - e_lo[u].q);
            r_z += z_c - (e_hi[u].zm - e_lo[u].zm);
            r_q += (e_d[u].sq - e_a[u].sq) - (e_hi[u].sq - e_lo[u].sq);
            float dc, ds;
            local_edge_trig_diff(e_d[u].cs, e_a[u].cs, dc, ds);
            c_c += dc;
            c_s += ds;
            spans_ok &= n_c < kEdgeSpanMax;
        }
    }
    // ---- the decision (k_local_flags_rows: S in [T - A / 2 -+ sqrt(N B) / 2], conditions (i) and (ii))
    const float hm = hn_pos - z_mid, fn = (float)c_n, fr = (float)r_n, fi = (float)(c_n - r_n);
    const float T = c_z - fn * hm, A = r_z - fr * hm;
    const float B = __builtin_fmaxf((r_q - 2.0f * hm * r_z) + fr * hm * hm, 0.0f) * 1.02f + 1e-3f * fr;
    const float mid = T - 0.5f * A, rad = 0.5005f * __builtin_amdgcn_sqrtf(fr * B);
    // (behind k_local_sums the table holds fixed-point sums: a candidate's z - L/2 rounded to 2^-16, its square to 2^-14, cos and
    // sin to 2^-9 — 5e-5 n_c on T, taken from the slack; 1e-4 N on B, inside its 1e-3 N; 2e-3 n_c < 0.004 n_c on the resultant,
    // counted among the unit vectors the two resultants may differ by)
    const float slack = 0.1f + fn * (1.05e-3f + 1e-6f * fn * Ln);
    const float r_c = __builtin_amdgcn_sqrtf(c_c * c_c + c_s * c_s);
    const float est_c = (local_atan2_fast(-c_s, -c_c) + 3.1415927f) * (Ln * 0.15915494f);
    const float shift_c = gm_min_image(hn_pos - est_c, Ln, bad);
    const float e_c = fr + 1.0f + 0.004f * fn;
    const float xr = e_c * __builtin_amdgcn_rcpf(r_c);                      // asin(x) <= x + (pi / 2 - 1) x^3 on [0, 1]
    const float emargin = 1e-4f * Ln + (xr + 0.5708f * xr * xr * xr) * (0.15916f * Ln);
    const bool same_image = e_c < r_c && ulo_g + shift_c > -halfn + emargin && uhi_g + shift_c < halfn - emargin;
    const bool head_near = __builtin_fabsf(mid) + rad < (halfn - 1e-3f * Ln) * fi;
    centre_above = mid > 0.0f;         // S > 0: the centre lies above the head, d = z_head - centre < 0
    // (prune == 2, GORDER_HIP_LOCAL_DECIDE_NOTHING: a measuring aid — what a membrane costs whose heads the bound cannot decide)
    return !redo && spans_ok && c_n > r_n && same_image && head_near && a.prune != 2 &&
           __builtin_fabsf(mid) > rad + slack;                              // (NaN anywhere: not decided)
}
template <typename EdgePtr>
__device__ __forceinline__ void local_decide_frame(const LocalArgs &a, const DecideFrame &F, uint32_t s, uint32_t f, EdgePtr edge) {
    int bad = 0;
    uint32_t n_open = 0;
    for (uint32_t m0 = threadIdx.x & ~63u; m0 < a.n_mol_total; m0 += 1024u) {         // (uniform per wave)
        const uint32_t m_raw = m0 + (threadIdx.x & 63u);
        const bool head_ok = m_raw < a.n_mol_total;
        const uint32_t m = head_ok ? m_raw : a.n_mol_total - 1u;          // idle lanes shadow the last head, write nothing
        bool centre_above;
        const bool decided = local_decide_head(a, F, f, m, edge, centre_above, bad);
        if (decided && head_ok)
            a.aflags[(size_t)(a.row0 + s) * a.n_mol_total + m] = (uint8_t)((centre_above ? 1 : 0) ^ (a.flip ? 1 : 0));
        n_open += (uint32_t)__popcll(__ballot(!decided && head_ok));
    }
    if (n_open != 0u && (threadIdx.x & 63u) == 0u) { atomicAdd(&a.need[s], n_open); a.need[a.n_slab] = 1u; }
    if (bad) raise_box_range(a.err, f);
}
__global__ __launch_bounds__(1024) void k_local_decide(LocalArgs a) {
    extern __shared__ LocalEdge l_edge[];
    const uint32_t s = blockIdx.x;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    const uint4 g = a.grid[s];
    const uint4 fk = reinterpret_cast<const uint4 *>(a.finfo)[s];
    DecideFrame F;
    F.nca = g.x; F.ncb = g.y; F.ka = g.z; F.kb = g.w; F.ncs = g.y + 2u * g.w; F.n_rows = 2u * g.z + 1u;
    F.da = (int)((a.dim + 1u) % 3u); F.db = (int)((a.dim + 2u) % 3u); F.dn = (int)a.dim;
    F.La = box[F.da]; F.Lb = box[F.db]; F.Ln = box[F.dn];
    F.z_min = local_key_float(fk.x); F.z_max = local_key_float(fk.y);
    // (uniform) frames the bound does not apply to — the conditions of k_local_flags_rows — and the frame whose distances are wanted
    const bool fail = !(F.ka >= 1u && F.kb >= 1u && F.n_rows <= 16u && fk.z == 0u && fk.x <= fk.y);
    if (a.sums && s == 0u && threadIdx.x == 0u) a.todo[0] = make_uint2(0u, 0u);                 // this slab's list starts empty
    if (fail || !(F.z_max - F.z_min < 0.75f * F.Ln) || (int)s == a.write_dist_frame || (a.sums && a.need[s] != 0u)) {
        if (threadIdx.x == 0u) { atomicAdd(&a.need[s], 1u); a.need[a.n_slab] = 1u; }
        return;
    }
    const LocalEdge *edge = a.edge + (size_t)s * (kLocalMaxCells1D * (kLocalMaxCells1D + 1u));
    const uint32_t n_entries = F.nca * (F.ncs + 1u);
    if (n_entries <= kDecideEntries) {                                      // (uniform)
        const uint4 *src = reinterpret_cast<const uint4 *>(edge);
        uint4 *dst = reinterpret_cast<uint4 *>(l_edge);
        for (uint32_t i = threadIdx.x; i < n_entries; i += 1024u) dst[i] = src[i];
        __syncthreads();
        local_decide_frame<const LocalEdge *>(a, F, s, f, l_edge);
    } else {
        local_decide_frame<const LocalEdge *>(a, F, s, f, edge);
    }
}

// The general passes for the heads k_local_flags_rows listed; one wave per head, a fixed grid walks the list.
__global__ __launch_bounds__(256) void k_local_flags_todo(LocalArgs a) {
    // how k_local_decide fared, for the host to choose the next submit's kernels by (see run_leaflets)
    if (a.need && a.summary && blockIdx.x == 0u && threadIdx.x < 64u) {
        uint32_t c = 0;
        for (uint32_t s = threadIdx.x; s < a.n_slab; s += 64u) c += a.need[s] != 0u ? 1u : 0u;
        c = wave_total_u32(c);
        if (threadIdx.x == 0u) {
            const uint32_t open = atomicAdd(&a.summary[0], c) + c, seen = atomicAdd(&a.summary[1], a.n_slab) + a.n_slab;
            if (a.summary_host) {
                a.summary_host[0] = open;
                a.summary_host[1] = seen;
                a.summary[0] = a.summary[1] = 0u;
            }
        }
    }
    const uint32_t n = a.todo[0].x;
    for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n; i += gridDim.x * 4u) {
        const uint2 e = a.todo[1u + i];
        local_flags_head(a, (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y));
    }
}

}  // namespace
