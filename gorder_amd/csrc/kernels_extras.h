// kernels_extras.h — scatter modes: ordermaps, timewise rows, geometry filter, united-atom hydrogens (k_bonds_extras, k_ua_extras, k_map_accumulate).
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

// =============================================================================================
// "Extras" kernels: ordermaps (ordermap.rs:100-113), timewise partial sums (timewise.rs:130-186,
// 277-283) and the united-atom path (uaorder.rs:375-437, 947-1104).  These modes are bound by their
// scatter atomics, not by the coordinate stream, so they use a plain structure: a thread owns one
// sample (or one united-atom carbon), gathers its atoms straight from global memory and walks the
// frames of its chunk one by one.  The main accumulators are kept in registers exactly like K1.
// =============================================================================================
struct ExtraArgs {
    int maps;                        // ordermaps on
    uint32_t plane;                  // 0 xy, 1 xz, 2 yz -> (z, y)   (input/ordermap.rs:44-50)
    float x0, y0, binx, biny;
    uint32_t nx, ny;
    unsigned long long *map_packed;  // [leaflets ? 2 : 1][n_acc][nx*ny] packed (count << 42) + sum, see k_fold_maps
    // sample staging (k_map_accumulate), or null: 64-bit words (plane-tile << 32 | tick), kMapNoSample where a lane has
    // no sample.  The `n` lanes of a tile that hold the molecules of one slot (a MapRun, from lane tid0) own contiguous
    // pieces.  United atoms (K = 3 hydrogens): the words [((tile * K + k) * kBlock + tid0) * rec_stride ...) laid out
    // [frame - rec_frame0][n].  Bond tiles block the frames by kRecFrames = 4: a tile's words of one block of frames are
    // [frame block][lane][frame in block] — a thread writes its own four samples of a stage, 32 bytes, no transposition —
    // and a run's piece of a frame block the kRecFrames * n words from kRecFrames * tid0 (128 bytes for the four molecules
    // a tile of the 256-lipid membrane holds per slot: whole cache lines for the reader).
    // (Five-byte samples — a 32-bit word and a byte in a second array — were tried: the producer gained 20 us per 3000
    // frames, the reader lost 50: its pieces fell below the 128-byte line)
    unsigned long long *map_rec;
    const uint32_t *item_run;        // per item: (tid0 << 16) | n
    uint32_t rec_frame0, rec_stride; // rec_stride: words per lane = frames of the sub-range rounded up to 16
    const float4 *dyn;               // dynamic membrane normals [n_frames][n_mol_total] (nx, ny, nz, cloud size) or null
    int bin_core;                    // both bin widths in [2^-40, 2^40]: grid_index may use the division core
    float inv_binx, inv_biny;        // 1 / bin (IEEE, from the host): GORDER_FLAG_UA_FAST_NORMALISE's tile index
    int axis;                        // the static normal is this coordinate axis (0..2), or -1
    int tw;                          // timewise on
    unsigned long long *tw_sums;     // [rows][3][n_acc]
    unsigned long long *tw_cnts;     // [rows][3][n_acc]
    unsigned long long tw_row0;      // row of this batch's frame 0
    // united atoms: sin/cos of the construction angles, evaluated on the host with libm like the reference
    float sin_tet, cos_tet, sin_ch3, cos_ch3, sin_half, cos_half;
    // geometry selection (geometry.rs): per-frame shapes [n_frames][8] = anchor xyz, extents xyz, radius, height
    int geom_kind, geom_invert, geom_orient;
    float geom_thr;                  // cylinder / sphere: local_radius_threshold(radius), d2 < geom_thr == sqrt(d2) < radius
    const float *shapes;
    // GORDER_FLAG_UA_FAST_NORMALISE: box edges and 1 / box edge per frame [n_frames][8] (k_inv_box: IEEE divisions, once per frame instead
    // of once per lane and frame), or null
    const float *inv_box;
};

// groan_rs Rectangular / Cylinder / Sphere ::inside (oracle: inside_shape), XOR invert (geometry.rs:181-189).
// No array is indexed by a run-time value (the cylinder's axis picks its operands by selects: arrays indexed by it went
// to scratch memory), and `sqrt(d2) < radius` is `d2 < e.geom_thr` (local_radius_threshold: the smallest float whose
// correctly rounded square root reaches the radius — the same samples, no IEEE square root per sample).
__device__ __forceinline__ bool geom_inside(const ExtraArgs &e, const float *sh, float px, float py, float pz,
                                            const float *box, bool pbc, int &bad) {
    bool in = true;
    if (e.geom_kind == GORDER_GEOM_CUBOID) {
        float x = px - sh[0], y = py - sh[1], z = pz - sh[2];
        if (pbc) {
            x = gm_wrap(x, box[0], bad); y = gm_wrap(y, box[1], bad); z = gm_wrap(z, box[2], bad);
            in = (x <= sh[3]) && (y <= sh[4]) && (z <= sh[5]);
        } else {
            in = (x >= 0.0f) && (x <= sh[3]) && (y >= 0.0f) && (y <= sh[4]) && (z >= 0.0f) && (z <= sh[5]);
        }
    } else if (e.geom_kind == GORDER_GEOM_CYLINDER) {
        // orientation o, in-plane axes a = (o + 1) % 3 and b = (o + 2) % 3
        const int o = e.geom_orient;
        const float dx = px - sh[0], dy = py - sh[1], dz = pz - sh[2];
        float x = o == 0 ? dx : (o == 1 ? dy : dz), da = o == 0 ? dy : (o == 1 ? dz : dx), db = o == 0 ? dz : (o == 1 ? dx : dy);
        if (pbc) {
            const float bo = o == 0 ? box[0] : (o == 1 ? box[1] : box[2]), ba = o == 0 ? box[1] : (o == 1 ? box[2] : box[0]),
                        bb = o == 0 ? box[2] : (o == 1 ? box[0] : box[1]);
            da = gm_min_image(da, ba, bad); db = gm_min_image(db, bb, bad); x = gm_wrap(x, bo, bad);
        }
        in = (da * da + db * db < e.geom_thr) && (pbc ? true : (x >= 0.0f)) && (x <= sh[7]);
    } else {
        float dx = px - sh[0], dy = py - sh[1], dz = pz - sh[2];
        if (pbc) { dx = gm_min_image(dx, box[0], bad); dy = gm_min_image(dy, box[1], bad); dz = gm_min_image(dz, box[2], bad); }
        in = (dx * dx + dy * dy) + dz * dz < e.geom_thr;
    }
    return in != (e.geom_invert != 0);
}

// BondLike::add_order for the scatter targets (bond.rs:184-215): maps and the per-frame LDS partials
template <bool STAGED_ONLY = false, bool NO_MAPS = false>
__device__ __forceinline__ void extras_add(const FrameArgs &a, const ExtraArgs &e, uint32_t gslot, uint32_t lslot,
                                           int tick, float px, float py, float pz, int leaflet /* -1 none */,
                                           int *l_tw, uint32_t *l_twn, uint32_t lstride,
                                           unsigned long long *rec = nullptr, bool skip_tw = false, bool fast_bin = false) {
    if (!NO_MAPS && (STAGED_ONLY || e.maps)) {
        float x, y;
        if (e.plane == 0) { x = px; y = py; }
        else if (e.plane == 1) { x = px; y = pz; }
        else { x = pz; y = py; }
        int ix, iy;
        if (fast_bin) {     // (united atoms with GORDER_FLAG_UA_FAST_NORMALISE, carbons the fast construction kept)
            ix = grid_index_fast(x, e.x0, e.inv_binx, e.nx);
            iy = grid_index_fast(y, e.y0, e.inv_biny, e.ny);
        } else {
            ix = grid_index(x, e.x0, e.binx, e.nx, e.bin_core != 0);
            iy = grid_index(y, e.y0, e.biny, e.ny, e.bin_core != 0);
        }
        if (ix >= 0 && iy >= 0) {
            // ONE atomic per sample: count and tick sum share a 64-bit word, and with leaflets only the
            // sample's own leaflet plane is touched (total = upper + lower, bond.rs:199-213); k_fold_maps
            // unpacks.  Scattered 64-bit atomics run at ~24 G/s on gfx950 whatever the scope or table size
            // (tools/microbench/atomic_scatter.hip), so their number is what counts.
            const size_t nt = (size_t)e.nx * e.ny, t = (size_t)ix * e.ny + (size_t)iy;
            if (STAGED_ONLY || rec) {   // staged: (plane * tiles + tile) << 32 | tick, added to the map by k_map_accumulate
                *rec = ((unsigned long long)((leaflet > 0 ? nt : 0) + t) << 32) | (unsigned long long)(uint32_t)tick;
            } else {
                const size_t w = leaflet > 0 ? a.n_acc : 0;
                atomicAdd(&e.map_packed[(w + gslot) * nt + t], kMapOne + (unsigned long long)(long long)tick);
            }
        }
    }
    if (!STAGED_ONLY && e.tw && !skip_tw) {
        atomicAdd(&l_tw[lslot], tick);
        atomicAdd(&l_twn[lslot], 1u);
        if (leaflet == 0) {         // (the lower leaflet's row is total - upper: gorder_hip_timewise takes the difference)
            atomicAdd(&l_tw[lstride + lslot], tick);
            atomicAdd(&l_twn[lstride + lslot], 1u);
        }
    }
}

// flush the block's per-frame partial sums to the timewise rows (one frame)
__device__ __forceinline__ void extras_flush_tw(const FrameArgs &a, const ExtraArgs &e, const uint32_t *slots,
                                                uint32_t n_slots, uint32_t f, int *l_tw, uint32_t *l_twn,
                                                uint32_t lstride) {
    for (uint32_t ls = threadIdx.x; ls < n_slots; ls += blockDim.x) {
        const size_t row = ((size_t)e.tw_row0 + f) * 3u * a.n_acc;
        for (uint32_t w = 0; w < 2; w++) {          // total, upper (the lower leaflet's row is their difference)
            const uint32_t n = l_twn[w * lstride + ls];
            if (n) {
                atomicAdd(&e.tw_sums[row + (size_t)w * a.n_acc + slots[ls]],
                          (unsigned long long)(long long)l_tw[w * lstride + ls]);
                atomicAdd(&e.tw_cnts[row + (size_t)w * a.n_acc + slots[ls]], (unsigned long long)n);
            }
            l_tw[w * lstride + ls] = 0;
            l_twn[w * lstride + ls] = 0;
        }
    }
}


// A thread's staged samples of one block of kRecFrames frames (from frame f_first; kMapNoSample where there is none) to
// the blocked layout described at ExtraArgs::map_rec: 32 bytes of its own.
__device__ __forceinline__ void rec_store(const ExtraArgs &e, uint32_t tile_id, uint32_t f_first, uint32_t tid,
                                          const unsigned long long (&v)[kRecFrames]) {
    static_assert(kRecFrames == 4, "a thread's four words");
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    ull2 *row = reinterpret_cast<ull2 *>(e.map_rec + (size_t)tile_id * kBlock * e.rec_stride +
                                         (size_t)((f_first - e.rec_frame0) / kRecFrames) * (kRecFrames * kBlock) + (size_t)tid * kRecFrames);
    __builtin_nontemporal_store(ull2{v[0], v[1]}, row);               // (written once, read once by k_map_accumulate)
    __builtin_nontemporal_store(ull2{v[2], v[3]}, row + 1);
}

// MAPS_ONLY: staged ordermap samples and nothing else (no geometry selection, timewise rows, per-molecule normals)
template <bool ACOS_COS, bool MAPS_ONLY>
__global__ __launch_bounds__(kBlock) void k_bonds_extras(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                          const float *__restrict__ box9,
                                                          const uint8_t *__restrict__ aflags,
                                                          const uint32_t *__restrict__ arow,
                                                          const Tile *__restrict__ tiles,
                                                          const Item *__restrict__ items,
                                                          const uint32_t *__restrict__ tile_slots, uint32_t n_tiles) {
    __shared__ unsigned long long l_s[2 * kBlock];
    __shared__ uint32_t l_n[2 * kBlock];
    __shared__ int l_tw[MAPS_ONLY ? 1 : 3 * kBlock];
    __shared__ uint32_t l_twn[MAPS_ONLY ? 1 : 3 * kBlock];
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const bool active = tid < t.n_items;
    Item it{0, 0, 0, 0, 0};
    if (active) it = items[t.item0 + tid];
    const uint32_t gslot = active ? tile_slots[t.slot0 + it.lslot] : 0;
    const uint32_t f_begin = a.frame0 + chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    const size_t fstride = (size_t)a.n_atoms * 3u;
    const float *pi = xyz + ((size_t)t.atom0 + it.li) * 3u;
    const float *pj = xyz + ((size_t)t.atom0 + it.lj) * 3u;
    if (!MAPS_ONLY)
        for (uint32_t k = tid; k < 3 * kBlock; k += kBlock) { l_tw[k] = 0; l_twn[k] = 0; }
    __syncthreads();
    SampleAcc acc;
    unsigned long long recs[kRecFrames] = {kMapNoSample, kMapNoSample, kMapNoSample, kMapNoSample};
    int bad = 0;
    // the two atoms of the next frames are fetched ahead of the arithmetic of this one (the gather is latency-bound)
    constexpr uint32_t kAhead = 2;
    float nx1[kAhead][3], nx2[kAhead][3];
#pragma unroll
    for (uint32_t u = 0; u < kAhead; u++) {
        const uint32_t fu = min(f_begin + u, a.n_frames - 1u);
#pragma unroll
        for (int d = 0; d < 3; d++) { nx1[u][d] = pi[(size_t)fu * fstride + d]; nx2[u][d] = pj[(size_t)fu * fstride + d]; }
    }
    for (uint32_t f = f_begin; f < f_end; f++) {
        unsigned long long rec = kMapNoSample;
        const float p1x = nx1[0][0], p1y = nx1[0][1], p1z = nx1[0][2];
        const float p2x = nx2[0][0], p2y = nx2[0][1], p2z = nx2[0][2];
#pragma unroll
        for (uint32_t u = 0; u + 1 < kAhead; u++)
#pragma unroll
            for (int d = 0; d < 3; d++) { nx1[u][d] = nx1[u + 1][d]; nx2[u][d] = nx2[u + 1][d]; }
        {
            const uint32_t fu = min(f + kAhead, a.n_frames - 1u);
#pragma unroll
            for (int d = 0; d < 3; d++) {
                nx1[kAhead - 1][d] = pi[(size_t)fu * fstride + d];
                nx2[kAhead - 1][d] = pj[(size_t)fu * fstride + d];
            }
        }
        if (active) {
            float vx = p2x - p1x, vy = p2y - p1y, vz = p2z - p1z;
            if (a.pbc) {
                const float *b = a.box9 + 9 * (size_t)f;
                vx = gm_min_image(vx, b[0], bad);
                vy = gm_min_image(vy, b[4], bad);
                vz = gm_min_image(vz, b[8], bad);
            }
            if (p1x != p1x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot, 1, it.mol, 0);
            else if (p2x != p2x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot, 1, it.mol, 1);
            // bond position = p1 + v / 2 (bond.rs:422); geometry filter (bond.rs:424-426)
            const float mx = p1x + vx / 2.0f, my = p1y + vy / 2.0f, mz = p1z + vz / 2.0f;
            bool in = true;
            if (!MAPS_ONLY && e.geom_kind) {
                float box[3] = {1.0f, 1.0f, 1.0f};
                if (a.pbc) { const float *b = a.box9 + 9 * (size_t)f; box[0] = b[0]; box[1] = b[4]; box[2] = b[8]; }
                in = geom_inside(e, e.shapes + 8 * (size_t)f, mx, my, mz, box, a.pbc != 0, bad);
            }
            if (in) {
                float sch;
                if (!MAPS_ONLY && e.dyn) {   // the molecule's own normal of this frame, fetched after the geometry test (bond.rs:429-431)
                    const float4 n = e.dyn[(size_t)f * a.n_mol_total + it.mol];
                    if (n.w < 3.0f) raise_error(a.err, GORDER_ERR_DYNAMIC_NORMAL, f, kStageTypes, gslot, 1, it.mol, (uint32_t)n.w);
                    const float n2sq = (n.x * n.x + n.y * n.y) + n.z * n.z;
                    sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, n.x, n.y, n.z, __builtin_sqrtf(n2sq), n2sq);
                } else if (!ACOS_COS && e.axis >= 0) {     // the static normal is a coordinate axis: K1's short form, same bits
                    bool rare = false;
                    sch = e.axis == 0 ? gm_sch_axis<0>(vx, vy, vz, rare) : (e.axis == 1 ? gm_sch_axis<1>(vx, vy, vz, rare) : gm_sch_axis<2>(vx, vy, vz, rare));
                    if (__builtin_expect(rare, 0)) sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
                } else {
                    sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
                }
                const int tick = gm_tick(sch);
                int leaflet = -1;
                if (a.leaflets) leaflet = a.aflags[(size_t)a.arow[f] * a.n_mol_total + it.mol] ? 1 : 0;
                acc.s_tot += tick;
                acc.n_tot += 1;
                if (leaflet == 0) { acc.s_up += tick; acc.n_up += 1; }
                extras_add<MAPS_ONLY>(a, e, gslot, it.lslot, tick, mx, my, mz, leaflet, l_tw, l_twn, kBlock,
                                      (MAPS_ONLY || e.map_rec) ? &rec : nullptr);
            }
        }
        if (MAPS_ONLY || e.map_rec) {   // staged samples: a thread's words of kRecFrames frames go out together
            const uint32_t c = (f - f_begin) % kRecFrames;      // the host keeps f_begin - rec_frame0 a multiple of kRecFrames
#pragma unroll
            for (uint32_t m = 0; m < kRecFrames; m++) recs[m] = m == c ? rec : recs[m];      // (selects: the array stays in registers)
            if (c == kRecFrames - 1 || f + 1 == f_end) {
                rec_store(e, tile_id, f - c, tid, recs);
#pragma unroll
                for (uint32_t m = 0; m < kRecFrames; m++) recs[m] = kMapNoSample;
            }
        }
        if (!MAPS_ONLY && e.tw) {
            __syncthreads();
            extras_flush_tw(a, e, tile_slots + t.slot0, t.n_slots, f, l_tw, l_twn, kBlock);
            __syncthreads();
        }
    }
    if (bad) raise_box_range(a.err, f_begin);
    l_s[tid] = 0; l_s[kBlock + tid] = 0; l_n[tid] = 0; l_n[kBlock + tid] = 0;
    __syncthreads();
    if (active && acc.n_tot) {
        atomicAdd(&l_s[it.lslot], (unsigned long long)acc.s_tot);
        atomicAdd(&l_n[it.lslot], acc.n_tot);
        if (acc.n_up) {
            atomicAdd(&l_s[kBlock + it.lslot], (unsigned long long)acc.s_up);
            atomicAdd(&l_n[kBlock + it.lslot], acc.n_up);
        }
    }
    __syncthreads();
    if (tid < t.n_slots && l_n[tid]) {
        unsigned long long *accp = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
        const uint32_t slot = tile_slots[t.slot0 + tid];
        atomicAdd(&accp[slot], l_s[tid]);
        atomicAdd(&accp[2u * a.n_acc + slot], (unsigned long long)l_n[tid]);
        if (l_n[kBlock + tid]) {
            atomicAdd(&accp[a.n_acc + slot], l_s[kBlock + tid]);
            atomicAdd(&accp[3u * a.n_acc + slot], (unsigned long long)l_n[kBlock + tid]);
        }
    }
}

// ---- ordermaps and nothing else, bonds: K1's staging (kernels_bonds.h: a tile's atom window of G = kRecFrames frames
// HBM -> registers -> LDS, every byte read once, the next stage's loads in flight) with the staged map words as a second
// output: compute_core<MAPS> leaves a thread's word per frame of the stage in registers, and they go out as one 16-byte
// store + one of four bytes (rec_store).  k_bonds_extras<*, true> — a thread gathers its two atoms per frame from global memory — took 294 us per
// 3000 frames of the 256-lipid all-atom membrane where this kernel's read-only twin takes 144.
// grid = n_tiles * n_chunks (frames_per_chunk a multiple of kRecFrames, a.frame0 = the sub-range's first frame).
template <int NPF, bool PBC, bool LEAF, int AXIS>
__global__ __launch_bounds__(kBlock, 4) void k_bonds_tiled_maps(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                               const float *__restrict__ box9, const uint8_t *__restrict__ aflags,
                                                               const uint32_t *__restrict__ arow, const Tile *__restrict__ tiles,
                                                               const Item *__restrict__ items, const uint32_t *__restrict__ tile_slots,
                                                               uint32_t n_tiles, uint32_t lw) {
    constexpr int G = (int)kRecFrames;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using S = TiledStage<G, NPF, false, PBC, LEAF, AXIS>;
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const uint32_t sk = tid / S::TPF, si = tid % S::TPF;
    const bool active = tid < t.n_items;
    Item it{0, 0, 0, 0, 0};
    if (active) it = items[t.item0 + tid];
    const uint32_t f_begin = a.frame0 + chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    const uint32_t f_full = f_begin + ((f_end - f_begin) / G) * G;
    unsigned long long words[kRecFrames] = {kMapNoSample, kMapNoSample, kMapNoSample, kMapNoSample};       // (padding lanes: none)
    TiledMapOut mo;
    mo.plane = e.plane; mo.x0 = e.x0; mo.y0 = e.y0; mo.binx = e.binx; mo.biny = e.biny; mo.nx = e.nx; mo.ny = e.ny;
    mo.bin_core = e.bin_core; mo.words = words;
    SampleAcc acc;
    int bad = 0;
    uint32_t nan_which = 0, nan_frame = kNoNan;
    v4f pre[NPF];
    if (f_begin < f_full) S::template load<false>(a, t, f_begin, f_end, sk, si, pre);
    for (uint32_t f0 = f_begin; f0 < f_full; f0 += G) {
        S::template store<false>(a, t, f0, f_end, sk, si, pre, lds, lw);
        __syncthreads();
        if (f0 + G < f_full) S::template load<false>(a, t, f0 + G, f_end, sk, si, pre);
        if (active) S::template compute<1>(a, t, it, f0, lds, lw, acc, bad, nan_which, nan_frame, &mo);
        rec_store(e, tile_id, f0, tid, words);
        __syncthreads();
    }
    if (f_full < f_end) {   // last, partial stage of the sub-range
        S::template load<true>(a, t, f_full, f_end, sk, si, pre);
        S::template store<true>(a, t, f_full, f_end, sk, si, pre, lds, lw);
        __syncthreads();
#pragma unroll
        for (uint32_t m = 0; m < kRecFrames; m++) words[m] = kMapNoSample;
        if (active) S::template compute_tail<1>(a, t, it, f_full, f_end, lds, lw, acc, bad, nan_which, nan_frame, &mo);
        rec_store(e, tile_id, f_full, tid, words);
        __syncthreads();
    }
    if (nan_frame != kNoNan)
        raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_frame, kStageTypes, tile_slots[t.slot0 + it.lslot], 1, it.mol, nan_which);
    if (bad) raise_box_range(a.err, f_begin);
    __syncthreads();
    unsigned long long *l_s = reinterpret_cast<unsigned long long *>(lds);   // [2][256]
    uint32_t *l_n = reinterpret_cast<uint32_t *>(l_s + 2 * kBlock);          // [2][256]
    l_s[tid] = 0; l_s[kBlock + tid] = 0; l_n[tid] = 0; l_n[kBlock + tid] = 0;
    __syncthreads();
    if (active && acc.n_tot) {
        atomicAdd(&l_s[it.lslot], (unsigned long long)acc.s_tot);
        atomicAdd(&l_n[it.lslot], acc.n_tot);
        if (acc.n_up) {
            atomicAdd(&l_s[kBlock + it.lslot], (unsigned long long)acc.s_up);
            atomicAdd(&l_n[kBlock + it.lslot], acc.n_up);
        }
    }
    __syncthreads();
    if (tid < t.n_slots && l_n[tid]) {
        unsigned long long *accp = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
        const uint32_t slot = tile_slots[t.slot0 + tid];
        atomicAdd(&accp[slot], l_s[tid]);
        atomicAdd(&accp[2u * a.n_acc + slot], (unsigned long long)l_n[tid]);
        if (l_n[kBlock + tid]) {
            atomicAdd(&accp[a.n_acc + slot], l_s[kBlock + tid]);
            atomicAdd(&accp[3u * a.n_acc + slot], (unsigned long long)l_n[kBlock + tid]);
        }
    }
}

// ---- per-frame rows (timewise sums, timewise.rs:130-186), alone or (MAPS) with staged ordermaps, bonds: K1's staging
// again, the stage's ticks as a second output (and the ordermap words as a third, written like k_bonds_tiled_maps').  The tile's items come in slot order (MapRun), so the samples of a slot are neighbouring lanes: a
// lane leaves its tick (and leaflet) of each frame of the stage in LDS, and one thread per slot of the tile adds its run
// up — no LDS atomics (k_bonds_extras: two to four per sample, one lane per cycle) — and sends the frame's partial sums
// to the rows with the same global atomics as extras_flush_tw.  No barrier of its own: the ticks are written before the
// barrier that ends a stage and read behind it, next to the staging of the next stage.
// grid = n_tiles * n_chunks; items = the slot-ordered copy, e.item_run its runs.
// MOM: as in k_bonds_tiled — the tile's share of the membrane group's normal coordinates summed on the way, for a batch whose
// leaflets are assigned from this kernel's own read of the frames.
template <int NPF, bool PBC, bool LEAF, int AXIS, bool MAPS, bool MOM = false>
__global__ __launch_bounds__(kBlock, 4) void k_bonds_tiled_tw(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                             const float *__restrict__ box9, const uint8_t *__restrict__ aflags,
                                                             const uint32_t *__restrict__ arow, const Tile *__restrict__ tiles,
                                                             const Item *__restrict__ items, const uint32_t *__restrict__ tile_slots,
                                                             uint32_t n_tiles, uint32_t lw) {
    constexpr int G = (int)kRecFrames;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int l_tick[kRecFrames][kBlock];
    __shared__ uint8_t l_side[kRecFrames][kBlock];      // 0 upper, 1 lower (LEAF), 2: no sample
    __shared__ uint32_t l_slot_run[kBlock];             // per slot of the tile: (first lane << 16) | lanes
    __shared__ uint32_t l_slot_id[kBlock];              // ... and its accumulator slot
    using S = TiledStage<G, NPF, false, PBC, LEAF, AXIS>;
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const uint32_t sk = tid / S::TPF, si = tid % S::TPF;
    const bool active = tid < t.n_items;
    Item it{0, 0, 0, 0, 0};
    if (active) {
        it = items[t.item0 + tid];
        const uint32_t run = e.item_run[t.item0 + tid];
        if ((run >> 16) == tid) l_slot_run[it.lslot] = run;          // the first lane of a run writes it down
    }
    const uint32_t f_begin = a.frame0 + chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    unsigned long long words[kRecFrames], mwords[MAPS ? kRecFrames : 1];
    TiledMapOut mo{};
    if (MAPS) {                     // the ordermap words as well (staged like k_bonds_tiled_maps')
        mo.plane = e.plane; mo.x0 = e.x0; mo.y0 = e.y0; mo.binx = e.binx; mo.biny = e.biny; mo.nx = e.nx; mo.ny = e.ny;
        mo.bin_core = e.bin_core; mo.words = mwords; mo.ticks = words;
    } else {
        mo.nx = 0;                  // ticks only: word = (lower << 32) | tick
        mo.words = words;
    }
    SampleAcc acc;
    int bad = 0;
    uint32_t nan_which = 0, nan_frame = kNoNan;
    v4f pre[NPF];
    const uint32_t my_slot = tid < t.n_slots ? tile_slots[t.slot0 + tid] : 0u;
    l_slot_id[tid] = my_slot;
    uint2 own = make_uint2(0u, 0u), my_head = make_uint2(0u, 0xffffffffu);
    if (MOM) {
        own = a.own[tile_id];
        const uint32_t hq = a.own_head_begin[tile_id] + (tid & 63u);
        if (hq < a.own_head_begin[tile_id + 1u]) my_head = a.own_heads[hq];
    }
    if (f_begin + G <= f_end) S::template load<false>(a, t, f_begin, f_end, sk, si, pre);
    for (uint32_t fs = f_begin; fs < f_end; fs += G) {
        const bool full = fs + G <= f_end;
        bool finite;
        if (full) {
            finite = S::template store<false, MOM>(a, t, fs, f_end, sk, si, pre, lds, lw);
        } else {
            S::template load<true>(a, t, fs, f_end, sk, si, pre);
            finite = S::template store<true, MOM>(a, t, fs, f_end, sk, si, pre, lds, lw);
        }
        __syncthreads();
        if (full && fs + 2u * G <= f_end) S::template load<false>(a, t, fs + G, f_end, sk, si, pre);     // next stage in flight
        if (MOM && fs + sk < f_end) {
            static_assert(!MOM || (uint32_t)G * 64u == kBlock, "a wave per frame slot");
            tiled_moments(a, t, own, tile_id, n_tiles, fs + sk, lds + (size_t)sk * lw, finite, my_head);
        }
#pragma unroll
        for (uint32_t k = 0; k < kRecFrames; k++) { words[k] = kMapNoSample; if (MAPS) mwords[k] = kMapNoSample; }
        if (active) {
            if (full) S::template compute<MAPS ? 3 : 2>(a, t, it, fs, lds, lw, acc, bad, nan_which, nan_frame, &mo);
            else S::template compute_tail<MAPS ? 3 : 2>(a, t, it, fs, f_end, lds, lw, acc, bad, nan_which, nan_frame, &mo);
        }
        if constexpr (MAPS) rec_store(e, tile_id, fs, tid, mwords);
#pragma unroll
        for (uint32_t k = 0; k < kRecFrames; k++) {
            l_tick[k][tid] = (int)(uint32_t)words[k];
            l_side[k][tid] = words[k] == kMapNoSample ? (uint8_t)2 : (uint8_t)(words[k] >> 32);
        }
        __syncthreads();
        // a thread per (slot of the tile, frame of the stage) adds the slot's run up; the next stage's staging goes on
        // meanwhile (other LDS), and its ticks are written behind its own barrier
        for (uint32_t w = tid; w < kRecFrames * t.n_slots; w += kBlock) {
            const uint32_t k = w % kRecFrames, ls = w / kRecFrames;
            if (fs + k >= f_end) continue;
            const uint32_t run = l_slot_run[ls], tid0 = run >> 16, n = run & 0xffffu;
            int s_all = 0, s_low = 0;
            uint32_t n_all = 0, n_low = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t side = l_side[k][tid0 + j];
                const int tick = l_tick[k][tid0 + j];
                if (side == 2u) continue;
                s_all += tick; n_all += 1u;
                if (side == 1u) { s_low += tick; n_low += 1u; }
            }
            if (!n_all) continue;
            const uint32_t slot = l_slot_id[ls];
            const size_t row = ((size_t)e.tw_row0 + fs + k) * 3u * a.n_acc;
            atomicAdd(&e.tw_sums[row + slot], (unsigned long long)(long long)s_all);
            atomicAdd(&e.tw_cnts[row + slot], (unsigned long long)n_all);
            if (LEAF) {
                const uint32_t n_up = n_all - n_low;
                if (n_up) {
                    atomicAdd(&e.tw_sums[row + (size_t)a.n_acc + slot], (unsigned long long)(long long)(s_all - s_low));
                    atomicAdd(&e.tw_cnts[row + (size_t)a.n_acc + slot], (unsigned long long)n_up);
                }
                // (no atomics for the lower leaflet: its row is total - upper, taken by gorder_hip_timewise — a third of this
                // kernel's row traffic)
            }
        }
    }
    __syncthreads();
    if (nan_frame != kNoNan)
        raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_frame, kStageTypes, tile_slots[t.slot0 + it.lslot], 1, it.mol, nan_which);
    if (bad) raise_box_range(a.err, f_begin);
    unsigned long long *l_s = reinterpret_cast<unsigned long long *>(lds);   // [2][256]
    uint32_t *l_n = reinterpret_cast<uint32_t *>(l_s + 2 * kBlock);          // [2][256]
    l_s[tid] = 0; l_s[kBlock + tid] = 0; l_n[tid] = 0; l_n[kBlock + tid] = 0;
    __syncthreads();
    if (active && acc.n_tot) {
        atomicAdd(&l_s[it.lslot], (unsigned long long)acc.s_tot);
        atomicAdd(&l_n[it.lslot], acc.n_tot);
        if (acc.n_up) {
            atomicAdd(&l_s[kBlock + it.lslot], (unsigned long long)acc.s_up);
            atomicAdd(&l_n[kBlock + it.lslot], acc.n_up);
        }
    }
    __syncthreads();
    if (tid < t.n_slots && l_n[tid]) {
        unsigned long long *accp = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
        atomicAdd(&accp[my_slot], l_s[tid]);
        atomicAdd(&accp[2u * a.n_acc + my_slot], (unsigned long long)l_n[tid]);
        if (l_n[kBlock + tid]) {
            atomicAdd(&accp[a.n_acc + my_slot], l_s[kBlock + tid]);
            atomicAdd(&accp[3u * a.n_acc + my_slot], (unsigned long long)l_n[kBlock + tid]);
        }
    }
}

// ---- united atoms: hydrogen construction, restating uaorder.rs:947-1104 with the operation order of
// nalgebra's Rotation3::from_axis_angle / matrix * vector and groan_rs' shift / wrap (oracle:
// gorder_oracle_predict_hydrogens).  All f32, no FMA.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3_cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float v3_norm(V3 a) { return __builtin_sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ V3 v3_unit(V3 a) { const float n = v3_norm(a); return {a.x / n, a.y / n, a.z / n}; }
__device__ __forceinline__ V3 v3_rotate(V3 u, float s, float c, V3 v) {
    const float sqx = u.x * u.x, sqy = u.y * u.y, sqz = u.z * u.z, omc = 1.0f - c;
    const float m11 = sqx + (1.0f - sqx) * c, m12 = u.x * u.y * omc - u.z * s, m13 = u.x * u.z * omc + u.y * s;
    const float m21 = u.x * u.y * omc + u.z * s, m22 = sqy + (1.0f - sqy) * c, m23 = u.y * u.z * omc - u.x * s;
    const float m31 = u.x * u.z * omc - u.y * s, m32 = u.y * u.z * omc + u.x * s, m33 = sqz + (1.0f - sqz) * c;
    return {(m11 * v.x + m12 * v.y) + m13 * v.z, (m21 * v.x + m22 * v.y) + m23 * v.z,
            (m31 * v.x + m32 * v.y) + m33 * v.z};
}
// Periodic-boundary policies for the hydrogen construction.  PbcStep does one select-only shift per
// operation and raises `slow` when that was not enough; PbcLoop is the literal `while` form of the
// reference.  The kernel evaluates a carbon with PbcStep and, only if `slow` came up, again with PbcLoop.
// (all state in scalars and every aggregate passed by value: nothing here may end up in scratch)
struct PbcStep {
    V3 box;
    bool pbc;
    bool slow = false;
    int bad = 0;
    __device__ __forceinline__ float len(int k) const { return k == 0 ? box.x : (k == 1 ? box.y : box.z); }
    __device__ __forceinline__ float mi(float d, int k) { return pbc ? gm_min_image_step(d, len(k), slow) : d; }
    __device__ __forceinline__ float wr(float x, int k) {
        if (!pbc) return x;
        const float L = len(k);
        const float r = x > L ? x - L : (x < 0.0f ? x + L : x);
        slow = slow || (r > L) || (r < 0.0f);
        return r;
    }
    // a / |a| with the cores of the IEEE square root and division (gm_sqrt_core, gm_div_core) and ONE reciprocal
    // refinement for the three quotients: the same bits as v3_unit while |a|^2 lies in [2^-40, 2^40] (a component
    // below 2^-103 aside, which no coordinate difference produces); anything else raises `slow`.
    __device__ __forceinline__ V3 unit(V3 a) {
        const float s2 = (a.x * a.x + a.y * a.y) + a.z * a.z;
        slow = slow || !(s2 >= 0x1p-40f && s2 <= 0x1p+40f);
        const float n = gm_sqrt_core(s2);
        float r = __builtin_amdgcn_rcpf(n);
        r = __builtin_fmaf(__builtin_fmaf(-n, r, 1.0f), r, r);
        auto quot = [&](float c) {
            float q = c * r;
            q = __builtin_fmaf(__builtin_fmaf(-n, q, c), r, q);
            return __builtin_fmaf(__builtin_fmaf(-n, q, c), r, q);
        };
        return {quot(a.x), quot(a.y), quot(a.z)};
    }
};
struct PbcLoop {
    V3 box;
    bool pbc;
    bool slow = false;
    int bad = 0;
    __device__ __forceinline__ float len(int k) const { return k == 0 ? box.x : (k == 1 ? box.y : box.z); }
    __device__ __forceinline__ float mi(float d, int k) { return pbc ? gm_min_image_loop(d, len(k), bad) : d; }
    __device__ __forceinline__ float wr(float x, int k) { return pbc ? gm_wrap(x, len(k), bad) : x; }
    __device__ __forceinline__ V3 unit(V3 a) { return v3_unit(a); }
};
template <typename PB>
__device__ __forceinline__ V3 v3_to(V3 p1, V3 p2, PB &pb) {
    return {pb.mi(p2.x - p1.x, 0), pb.mi(p2.y - p1.y, 1), pb.mi(p2.z - p1.z, 2)};
}
template <typename PB>
__device__ __forceinline__ V3 v3_shift_wrap(V3 t, V3 dir, PB &pb) {
    const V3 u = pb.unit(dir);
    return {pb.wr(t.x + u.x * 0.109f, 0), pb.wr(t.y + u.y * 0.109f, 1), pb.wr(t.z + u.z * 0.109f, 2)};   // BOND_LENGTH
}

struct UaConsts {
    float sin_tet, cos_tet, sin_ch3, cos_ch3, sin_half, cos_half;
};
struct UaCarbon {       // the carbon's atoms: helper1,target,helper2,- or h1,h2,h3,target (CH1 saturated)
    V3 p0, p1, p2, p3;
};
struct UaBonds {        // per hydrogen: the C->H vector and the bond position (unused entries are zero)
    V3 v0, v1, v2, b0, b1, b2;
    int bad;
};

// hydrogens of one united-atom carbon, then per hydrogen the C->H vector and the bond position
// (UAAtom::calculate_sch, uaorder.rs:375-397: vec = target -> H, position = H + vec / 2 (sic))
template <typename PB>
__device__ __forceinline__ UaBonds ua_carbon(uint32_t kind, UaCarbon c, UaConsts e, PB &pb) {
    const V3 zero{0.0f, 0.0f, 0.0f};
    V3 h0 = zero, h1 = zero, h2 = zero, target = c.p1;
    if (kind == GORDER_UA_CH3) {            // uaorder.rs:947-981
        const V3 th1 = v3_to(target, c.p0, pb), th2 = v3_to(target, c.p2, pb);
        const V3 ua = pb.unit(v3_cross(th2, th1));
        const V3 hv1 = v3_rotate(ua, e.sin_tet, e.cos_tet, th1);
        h0 = v3_shift_wrap(target, hv1, pb);
        const V3 n1 = pb.unit(th1);
        h1 = v3_shift_wrap(target, v3_rotate(n1, e.sin_ch3, e.cos_ch3, hv1), pb);
        h2 = v3_shift_wrap(target, v3_rotate(n1, -e.sin_ch3, e.cos_ch3, hv1), pb);
    } else if (kind == GORDER_UA_CH2) {     // uaorder.rs:985-1020
        const V3 th1 = pb.unit(v3_to(target, c.p0, pb)), th2 = pb.unit(v3_to(target, c.p2, pb));
        const V3 pn = v3_cross(th2, th1);
        const V3 ra = pb.unit(V3{th1.x - th2.x, th1.y - th2.y, th1.z - th2.z});
        const V3 rv = v3_cross(pn, ra);
        const V3 ura = pb.unit(ra);
        h0 = v3_shift_wrap(target, v3_rotate(ura, e.sin_half, e.cos_half, rv), pb);
        h1 = v3_shift_wrap(target, v3_rotate(ura, -e.sin_half, e.cos_half, rv), pb);
    } else if (kind == GORDER_UA_CH1_UNSAT) {   // uaorder.rs:1024-1045
        const V3 th1 = v3_to(target, c.p0, pb), th2 = v3_to(target, c.p2, pb);
        const float prod = (th1.x * th2.x + th1.y * th2.y) + th1.z * th2.z;
        const float n1 = v3_norm(th1), n2 = v3_norm(th2);
        float gamma = 0.0f;
        if (!(n1 == 0.0f || n2 == 0.0f)) {
            float cs = prod / (n1 * n2);
            cs = cs < -1.0f ? -1.0f : (cs > 1.0f ? 1.0f : cs);
            gamma = gm_acosf(cs);
        }
        const float ang = 3.14159265358979323846f - (gamma / 2.0f);
        // own sin / cos kernels (gm_math.h), restated by the oracle's non-libm modes: the hydrogen is the same bits there
        const float sn = gm_sinf_0pi(ang), cs = gm_cosf(ang);
        const V3 ua = pb.unit(v3_cross(th1, th2));
        h0 = v3_shift_wrap(target, ang == 0.0f ? th2 : v3_rotate(ua, sn, cs, th2), pb);
    } else {                                // CH1 saturated, uaorder.rs:1087-1104 (h1, h2, h3, target)
        target = c.p3;
        const V3 t1 = pb.unit(v3_to(target, c.p0, pb)), t2 = pb.unit(v3_to(target, c.p1, pb)),
                 t3 = pb.unit(v3_to(target, c.p2, pb));
        h0 = v3_shift_wrap(target, V3{-((t1.x + t2.x) + t3.x), -((t1.y + t2.y) + t3.y), -((t1.z + t2.z) + t3.z)}, pb);
    }
    UaBonds r;
    r.v0 = v3_to(target, h0, pb);
    r.b0 = {h0.x + r.v0.x / 2.0f, h0.y + r.v0.y / 2.0f, h0.z + r.v0.z / 2.0f};
    r.v1 = r.v2 = r.b1 = r.b2 = zero;
    if (kind == GORDER_UA_CH3 || kind == GORDER_UA_CH2) {
        r.v1 = v3_to(target, h1, pb);
        r.b1 = {h1.x + r.v1.x / 2.0f, h1.y + r.v1.y / 2.0f, h1.z + r.v1.z / 2.0f};
    }
    if (kind == GORDER_UA_CH3) {
        r.v2 = v3_to(target, h2, pb);
        r.b2 = {h2.x + r.v2.x / 2.0f, h2.y + r.v2.y / 2.0f, h2.z + r.v2.z / 2.0f};
    }
    r.bad = pb.bad;
    return r;
}
// ---- the same constructions with PAIRS of vectors in packed registers (v_pk_mul / v_pk_add / v_pk_fma_f32 do two
// f32 operations per lane and instruction).  A methylene carbon does everything twice — two helper vectors are made
// periodic and normalised, two hydrogens are rotated (by +- the same angle about the same axis), shifted, wrapped and
// turned into bond vectors — and a methyl carbon does so for its second and third hydrogen.  Lane 0 / lane 1 of an `f2`
// hold the two instances; every component goes through exactly the operations of the scalar code above (IEEE
// multiplication is sign-symmetric, so u * (-s) = -(u * s) and a - (-b) = a + b bit for bit), hence the same bits.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
struct V3P { f2 x, y, z; };
__device__ __forceinline__ f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 f2_splat(float v) { return f2{v, v}; }
__device__ __forceinline__ V3 v3p_lane0(V3P a) { return {a.x.x, a.y.x, a.z.x}; }
__device__ __forceinline__ V3 v3p_lane1(V3P a) { return {a.x.y, a.y.y, a.z.y}; }
struct PbcStep2 {       // PbcStep on pairs
    V3 box;
    bool pbc;
    i2 slow2 = {0, 0};       // the "one shift was not enough" conditions, OR-ed per lane; reduced once by slow()
    __device__ __forceinline__ bool slow() const { return (slow2.x | slow2.y) != 0; }
    __device__ __forceinline__ f2 mi(f2 d, float L) {
        if (!pbc) return d;
        const f2 Lv = f2_splat(L), half = Lv / 2.0f;
        const f2 r = __builtin_elementwise_abs(d) > half ? d - __builtin_elementwise_copysign(Lv, d) : d;
        slow2 |= __builtin_elementwise_abs(r) > half;
        return r;
    }
    __device__ __forceinline__ f2 wr(f2 x, float L) {
        if (!pbc) return x;
        const f2 Lv = f2_splat(L), zero = f2_splat(0.0f);
        const f2 r = x > Lv ? x - Lv : (x < zero ? x + Lv : x);
        slow2 |= (r > Lv) | (r < zero);
        return r;
    }
    __device__ __forceinline__ V3P unit(V3P a) {       // PbcStep::unit, twice
        const f2 s2 = (a.x * a.x + a.y * a.y) + a.z * a.z;
        slow2 |= ~((s2 >= f2_splat(0x1p-40f)) & (s2 <= f2_splat(0x1p+40f)));
        const f2 s = f2{__builtin_amdgcn_sqrtf(s2.x), __builtin_amdgcn_sqrtf(s2.y)};      // gm_sqrt_core
        const i2 si = __builtin_bit_cast(i2, s);
        const f2 sm = __builtin_bit_cast(f2, si - 1), sp = __builtin_bit_cast(f2, si + 1);
        const f2 rm = f2_fma(-sm, s, s2), rp = f2_fma(-sp, s, s2);
        f2 n = rm <= f2_splat(0.0f) ? sm : s;
        n = rp > f2_splat(0.0f) ? sp : n;
        f2 r = f2{__builtin_amdgcn_rcpf(n.x), __builtin_amdgcn_rcpf(n.y)};
        r = f2_fma(f2_fma(-n, r, f2_splat(1.0f)), r, r);
        auto quot = [&](f2 c) {
            f2 q = c * r;
            q = f2_fma(f2_fma(-n, q, c), r, q);
            return f2_fma(f2_fma(-n, q, c), r, q);
        };
        return {quot(a.x), quot(a.y), quot(a.z)};
    }
};
// target -> the two points (p, q), periodic
__device__ __forceinline__ V3P v3p_to(V3 t, V3 p, V3 q, PbcStep2 &pb) {
    return {pb.mi(f2{p.x, q.x} - f2_splat(t.x), pb.box.x), pb.mi(f2{p.y, q.y} - f2_splat(t.y), pb.box.y),
            pb.mi(f2{p.z, q.z} - f2_splat(t.z), pb.box.z)};
}
__device__ __forceinline__ V3P v3p_to(V3 t, V3P h, PbcStep2 &pb) {
    return {pb.mi(h.x - f2_splat(t.x), pb.box.x), pb.mi(h.y - f2_splat(t.y), pb.box.y), pb.mi(h.z - f2_splat(t.z), pb.box.z)};
}
// v rotated about the unit axis u by +angle (lane 0) and -angle (lane 1): v3_rotate with s = (+s, -s)
__device__ __forceinline__ V3P v3p_rotate_pm(V3 u, float s, float c, V3 v) {
    const float sqx = u.x * u.x, sqy = u.y * u.y, sqz = u.z * u.z, omc = 1.0f - c;
    const f2 sv = f2{s, -s};
    const f2 uxs = f2_splat(u.x) * sv, uys = f2_splat(u.y) * sv, uzs = f2_splat(u.z) * sv;
    const float xy = u.x * u.y * omc, xz = u.x * u.z * omc, yz = u.y * u.z * omc;
    const float m11 = sqx + (1.0f - sqx) * c, m22 = sqy + (1.0f - sqy) * c, m33 = sqz + (1.0f - sqz) * c;
    const f2 m12 = f2_splat(xy) - uzs, m13 = f2_splat(xz) + uys;
    const f2 m21 = f2_splat(xy) + uzs, m23 = f2_splat(yz) - uxs;
    const f2 m31 = f2_splat(xz) - uys, m32 = f2_splat(yz) + uxs;
    const f2 vx = f2_splat(v.x), vy = f2_splat(v.y), vz = f2_splat(v.z);
    return {(f2_splat(m11) * vx + m12 * vy) + m13 * vz, (m21 * vx + f2_splat(m22) * vy) + m23 * vz,
            (m31 * vx + m32 * vy) + f2_splat(m33) * vz};
}
__device__ __forceinline__ V3P v3p_shift_wrap(V3 t, V3P dir, PbcStep2 &pb) {
    const V3P u = pb.unit(dir);
    return {pb.wr(f2_splat(t.x) + u.x * 0.109f, pb.box.x), pb.wr(f2_splat(t.y) + u.y * 0.109f, pb.box.y),
            pb.wr(f2_splat(t.z) + u.z * 0.109f, pb.box.z)};   // BOND_LENGTH
}
// CH2 and CH3 carbons by pairs; `slow` comes back raised when a literal-loop evaluation is needed instead
__device__ __forceinline__ UaBonds ua_carbon_pairs(uint32_t kind, UaCarbon c, UaConsts e, V3 box, bool pbc, bool &slow) {
    const V3 zero{0.0f, 0.0f, 0.0f};
    const V3 target = c.p1;
    PbcStep ps{box, pbc};
    PbcStep2 pp{box, pbc};
    UaBonds r;
    r.v2 = r.b2 = zero;
    V3P h;                                   // CH2: hydrogens 0, 1; CH3: hydrogens 1, 2
    if (kind == GORDER_UA_CH2) {            // uaorder.rs:985-1020
        const V3P th = pp.unit(v3p_to(target, c.p0, c.p2, pp));
        const V3 th1 = v3p_lane0(th), th2 = v3p_lane1(th);
        const V3 pn = v3_cross(th2, th1);
        const V3 ra = ps.unit(V3{th1.x - th2.x, th1.y - th2.y, th1.z - th2.z});
        const V3 rv = v3_cross(pn, ra);
        const V3 ura = ps.unit(ra);
        h = v3p_shift_wrap(target, v3p_rotate_pm(ura, e.sin_half, e.cos_half, rv), pp);
    } else {                                // CH3, uaorder.rs:947-981
        const V3P th = v3p_to(target, c.p0, c.p2, pp);
        const V3 th1 = v3p_lane0(th), th2 = v3p_lane1(th);
        const V3 ua = ps.unit(v3_cross(th2, th1));
        const V3 hv1 = v3_rotate(ua, e.sin_tet, e.cos_tet, th1);
        const V3 h0 = v3_shift_wrap(target, hv1, ps);
        const V3 n1 = ps.unit(th1);
        h = v3p_shift_wrap(target, v3p_rotate_pm(n1, e.sin_ch3, e.cos_ch3, hv1), pp);
        r.v0 = v3_to(target, h0, ps);
        r.b0 = {h0.x + r.v0.x / 2.0f, h0.y + r.v0.y / 2.0f, h0.z + r.v0.z / 2.0f};
    }
    // bond vectors target -> H and bond positions H + v / 2 (UAAtom::calculate_sch, uaorder.rs:375-397)
    const V3P v = v3p_to(target, h, pp);
    const V3P b = {h.x + v.x / 2.0f, h.y + v.y / 2.0f, h.z + v.z / 2.0f};
    if (kind == GORDER_UA_CH2) {
        r.v0 = v3p_lane0(v); r.b0 = v3p_lane0(b);
        r.v1 = v3p_lane1(v); r.b1 = v3p_lane1(b);
    } else {
        r.v1 = v3p_lane0(v); r.b1 = v3p_lane0(b);
        r.v2 = v3p_lane1(v); r.b2 = v3p_lane1(b);
    }
    r.bad = 0;
    slow = ps.slow || pp.slow();
    return r;
}

// ---- GORDER_FLAG_UA_FAST_NORMALISE: the same constructions with tolerance-bounded arithmetic ---------------------------
// Opt-in (include/gorder_hip.h).  The exact path above spends most of its instructions on being the reference's bits:
// six normalisations by correctly rounded square root and division (46 instructions each as pairs), select chains for
// the periodic shifts.  Here
//  * a / |a| is a * rsqrt(|a|^2) with |a|^2 by fused multiply-adds and rsqrt by an integer seed and three Newton steps
//    (every step an IEEE mul / fma: the oracle's FAST mode restates it operation for operation, so device and oracle sums
//    stay EQUAL; relative error ~1e-7, the reference's own two roundings leave 6e-8);
//  * the axis that is already a unit vector is not normalised again (Unit::new_normalize(rot_axis), uaorder.rs:990-1003);
//  * the hydrogen is target + dir * (rsqrt * 0.109) by one fma per component;
//  * minimum image and wrap are d - L * rint(d / L) and x - L * floor(x / L) with 1 / L from k_inv_box: for shifts of
//    at most one box length the same values as the reference's loops except AT the boundaries (|d| = L / 2, x = L) and
//    where d / L rounds across one; more than one box length (|k| > 1) goes to the literal-loop evaluation like before;
//  * rotations by Rodrigues' formula, v c + (u x v) s [+ u (u.v)(1 - c)], instead of nalgebra's rotation matrix — the
//    axis is perpendicular to the rotated vector by construction in all but the second methyl rotation;
//  * the C -> H vector is taken from the hydrogen BEFORE it is wrapped, (target + dir r) - target, not as the minimum
//    image of the wrapped hydrogen (the same vector; the reference's form loses the last bits of it when the hydrogen
//    lands across a box face), and the wrapped hydrogen — consumed only through the bond position of ordermaps and
//    geometry selections — is not computed at all when neither is on (POS = false).
// What it costs in fidelity is measured, not assumed: tools/ua_fast_fidelity.py (fraction of samples that move by a tick
// against the libm oracle, fraction of bond positions that change ordermap tile) -> profiles/r04_ua_fast_fidelity.json.
__device__ __forceinline__ float ua_fast_rsqrt(float x) {
    float y = __int_as_float(0x5f375a86 - (__float_as_int(x) >> 1));
    const float hx = 0.5f * x;
    y = y * __builtin_fmaf(-(hx * y), y, 1.5f);
    y = y * __builtin_fmaf(-(hx * y), y, 1.5f);
    y = y * __builtin_fmaf(-(hx * y), y, 1.5f);
    return y;
}
struct PbcFast {
    V3 box, inv;
    bool pbc;
    bool slow = false;
    __device__ __forceinline__ float mi1(float d, float L, float iL) {
        if (!pbc) return d;
        const float k = __builtin_rintf(d * iL);
        slow = slow || (__builtin_fabsf(k) > 1.0f);
        return __builtin_fmaf(-L, k, d);
    }
    __device__ __forceinline__ float wr1(float x, float L, float iL) {
        if (!pbc) return x;
        const float k = __builtin_floorf(x * iL);
        slow = slow || (__builtin_fabsf(k) > 1.0f);
        return __builtin_fmaf(-L, k, x);
    }
    __device__ __forceinline__ V3 to(V3 p1, V3 p2) {
        return {mi1(p2.x - p1.x, box.x, inv.x), mi1(p2.y - p1.y, box.y, inv.y), mi1(p2.z - p1.z, box.z, inv.z)};
    }
    // 1 / |a|; |a|^2 outside [2^-40, 2^40] (zero, inf, NaN included) raises `slow`
    __device__ __forceinline__ float rnorm(V3 a) {
        const float s2 = __builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x));
        slow = slow || !(s2 >= 0x1p-40f && s2 <= 0x1p+40f);
        return ua_fast_rsqrt(s2);
    }
    __device__ __forceinline__ V3 unit(V3 a) { const float r = rnorm(a); return {a.x * r, a.y * r, a.z * r}; }
    // target + dir / |dir| * BOND_LENGTH, NOT wrapped
    __device__ __forceinline__ V3 shift(V3 t, V3 dir) {
        const float r = rnorm(dir) * 0.109f;
        return {__builtin_fmaf(dir.x, r, t.x), __builtin_fmaf(dir.y, r, t.y), __builtin_fmaf(dir.z, r, t.z)};
    }
    __device__ __forceinline__ V3 wrap(V3 h) { return {wr1(h.x, box.x, inv.x), wr1(h.y, box.y, inv.y), wr1(h.z, box.z, inv.z)}; }
};
// v rotated about the unit axis u perpendicular to it: v c + (u x v) s
__device__ __forceinline__ V3 v3_rotate_perp(V3 u, float s, float c, V3 v) {
    const V3 w = v3_cross(u, v);
    return {__builtin_fmaf(w.x, s, v.x * c), __builtin_fmaf(w.y, s, v.y * c), __builtin_fmaf(w.z, s, v.z * c)};
}
// ... by +angle (lane 0) and -angle (lane 1)
__device__ __forceinline__ V3P v3p_rotate_perp_pm(V3 u, float s, float c, V3 v) {
    const V3 w = v3_cross(u, v);
    const f2 sv = f2{s, -s};
    return {f2_fma(f2_splat(w.x), sv, f2_splat(v.x * c)), f2_fma(f2_splat(w.y), sv, f2_splat(v.y * c)),
            f2_fma(f2_splat(w.z), sv, f2_splat(v.z * c))};
}
// ... about any unit axis: v c + (u x v) s + u (u.v)(1 - c)
__device__ __forceinline__ V3P v3p_rotate_rod_pm(V3 u, float s, float c, V3 v) {
    const V3 w = v3_cross(u, v);
    const float k = __builtin_fmaf(u.z, v.z, __builtin_fmaf(u.y, v.y, u.x * v.x)) * (1.0f - c);
    const f2 sv = f2{s, -s};
    return {f2_fma(f2_splat(u.x), f2_splat(k), f2_fma(f2_splat(w.x), sv, f2_splat(v.x * c))),
            f2_fma(f2_splat(u.y), f2_splat(k), f2_fma(f2_splat(w.y), sv, f2_splat(v.y * c))),
            f2_fma(f2_splat(u.z), f2_splat(k), f2_fma(f2_splat(w.z), sv, f2_splat(v.z * c)))};
}
struct PbcFast2 {       // PbcFast on pairs
    V3 box, inv;
    bool pbc;
    i2 slow2 = {0, 0};
    __device__ __forceinline__ bool slow() const { return (slow2.x | slow2.y) != 0; }
    __device__ __forceinline__ f2 mi1(f2 d, float L, float iL) {
        if (!pbc) return d;
        const f2 q = d * iL;
        const f2 k = f2{__builtin_rintf(q.x), __builtin_rintf(q.y)};
        slow2 |= __builtin_elementwise_abs(k) > f2_splat(1.0f);
        return f2_fma(f2_splat(-L), k, d);
    }
    __device__ __forceinline__ f2 wr1(f2 x, float L, float iL) {
        if (!pbc) return x;
        const f2 q = x * iL;
        const f2 k = f2{__builtin_floorf(q.x), __builtin_floorf(q.y)};
        slow2 |= __builtin_elementwise_abs(k) > f2_splat(1.0f);
        return f2_fma(f2_splat(-L), k, x);
    }
    __device__ __forceinline__ V3P to(V3 t, V3 p, V3 q) {
        return {mi1(f2{p.x, q.x} - f2_splat(t.x), box.x, inv.x), mi1(f2{p.y, q.y} - f2_splat(t.y), box.y, inv.y),
                mi1(f2{p.z, q.z} - f2_splat(t.z), box.z, inv.z)};
    }
    __device__ __forceinline__ V3P to(V3 t, V3P h) {
        return {mi1(h.x - f2_splat(t.x), box.x, inv.x), mi1(h.y - f2_splat(t.y), box.y, inv.y), mi1(h.z - f2_splat(t.z), box.z, inv.z)};
    }
    __device__ __forceinline__ f2 rnorm(V3P a) {
        const f2 s2 = f2_fma(a.z, a.z, f2_fma(a.y, a.y, a.x * a.x));
        slow2 |= ~((s2 >= f2_splat(0x1p-40f)) & (s2 <= f2_splat(0x1p+40f)));
        const i2 seed = i2{0x5f375a86, 0x5f375a86} - (__builtin_bit_cast(i2, s2) >> 1);
        f2 y = __builtin_bit_cast(f2, seed);
        const f2 hx = s2 * 0.5f;
        y = y * f2_fma(-(hx * y), y, f2_splat(1.5f));
        y = y * f2_fma(-(hx * y), y, f2_splat(1.5f));
        y = y * f2_fma(-(hx * y), y, f2_splat(1.5f));
        return y;
    }
    __device__ __forceinline__ V3P unit(V3P a) { const f2 r = rnorm(a); return {a.x * r, a.y * r, a.z * r}; }
    __device__ __forceinline__ V3P shift(V3 t, V3P dir) {
        const f2 r = rnorm(dir) * 0.109f;
        return {f2_fma(dir.x, r, f2_splat(t.x)), f2_fma(dir.y, r, f2_splat(t.y)), f2_fma(dir.z, r, f2_splat(t.z))};
    }
    __device__ __forceinline__ V3P wrap(V3P h) { return {wr1(h.x, box.x, inv.x), wr1(h.y, box.y, inv.y), wr1(h.z, box.z, inv.z)}; }
};
// every kind of carbon with the fast forms; `slow` comes back raised when the literal-loop evaluation is needed instead.
// POS: the bond positions (hydrogen wrapped into the box + half the bond vector) are consumed — ordermaps, geometry.
// (need_pos: the same at run time — the general kernel is compiled with POS and asks its arguments)
template <bool POS>
__device__ __forceinline__ UaBonds ua_carbon_fast(uint32_t kind, UaCarbon c, UaConsts e, V3 box, V3 inv, bool pbc, bool need_pos, bool &slow) {
    const V3 zero{0.0f, 0.0f, 0.0f};
    PbcFast ps{box, inv, pbc};
    PbcFast2 pp{box, inv, pbc};
    UaBonds r;
    r.v0 = r.v1 = r.v2 = r.b0 = r.b1 = r.b2 = zero;
    r.bad = 0;
    // C -> H from the unwrapped hydrogen; position = wrapped H + v / 2.  (Values in, values out: with `V3 &` parameters the
    // results of the branches below met as a phi of POINTERS and stayed in scratch.)
    auto bond_v = [&](V3 target, V3 hu) { return V3{hu.x - target.x, hu.y - target.y, hu.z - target.z}; };
    auto bond_b = [&](V3 hu, V3 v) {
        if (!(POS && need_pos)) return zero;
        const V3 h = ps.wrap(hu);
        return V3{h.x + v.x / 2.0f, h.y + v.y / 2.0f, h.z + v.z / 2.0f};
    };
    if (kind == GORDER_UA_CH2 || kind == GORDER_UA_CH3) {
        const V3 target = c.p1;
        V3P hu;                                  // CH2: hydrogens 0, 1; CH3: hydrogens 1, 2 (not wrapped)
        if (kind == GORDER_UA_CH2) {            // uaorder.rs:985-1020
            const V3P th = pp.unit(pp.to(target, c.p0, c.p2));
            const V3 th1 = v3p_lane0(th), th2 = v3p_lane1(th);
            const V3 pn = v3_cross(th2, th1);
            const V3 ra = ps.unit(V3{th1.x - th2.x, th1.y - th2.y, th1.z - th2.z});      // (not normalised a second time)
            const V3 rv = v3_cross(pn, ra);                                                // perpendicular to ra
            hu = pp.shift(target, v3p_rotate_perp_pm(ra, e.sin_half, e.cos_half, rv));
        } else {                                // CH3, uaorder.rs:947-981
            const V3P th = pp.to(target, c.p0, c.p2);
            const V3 th1 = v3p_lane0(th), th2 = v3p_lane1(th);
            const V3 ua = ps.unit(v3_cross(th2, th1));                                     // perpendicular to th1
            const V3 hv1 = v3_rotate_perp(ua, e.sin_tet, e.cos_tet, th1);
            const V3 h0 = ps.shift(target, hv1);
            r.v0 = bond_v(target, h0);
            r.b0 = bond_b(h0, r.v0);
            const V3 n1 = ps.unit(th1);
            hu = pp.shift(target, v3p_rotate_rod_pm(n1, e.sin_ch3, e.cos_ch3, hv1));
        }
        const V3P v = {hu.x - f2_splat(target.x), hu.y - f2_splat(target.y), hu.z - f2_splat(target.z)};
        V3P b = {f2_splat(0.0f), f2_splat(0.0f), f2_splat(0.0f)};
        if (POS && need_pos) { const V3P h = pp.wrap(hu); b = {h.x + v.x / 2.0f, h.y + v.y / 2.0f, h.z + v.z / 2.0f}; }
        // The pair is hydrogens (0, 1) of a methylene and (1, 2) of a methyl carbon.  As selects of VALUES: assigning
        // `r.v0, r.v1` in one branch and `r.v1, r.v2` in the other made the compiler keep the result in scratch and index
        // it by the kind (two scratch stores and loads per frame, and a wait for every load in flight behind them).
        const bool ch2 = kind == GORDER_UA_CH2;
        const V3 v_lo = v3p_lane0(v), v_hi = v3p_lane1(v), b_lo = v3p_lane0(b), b_hi = v3p_lane1(b);
        r.v0 = V3{ch2 ? v_lo.x : r.v0.x, ch2 ? v_lo.y : r.v0.y, ch2 ? v_lo.z : r.v0.z};
        r.b0 = V3{ch2 ? b_lo.x : r.b0.x, ch2 ? b_lo.y : r.b0.y, ch2 ? b_lo.z : r.b0.z};
        r.v1 = V3{ch2 ? v_hi.x : v_lo.x, ch2 ? v_hi.y : v_lo.y, ch2 ? v_hi.z : v_lo.z};
        r.b1 = V3{ch2 ? b_hi.x : b_lo.x, ch2 ? b_hi.y : b_lo.y, ch2 ? b_hi.z : b_lo.z};
        r.v2 = v_hi; r.b2 = b_hi;                  // (a methylene carbon has no third hydrogen: never read)
    } else {
        V3 target = c.p1, hu;
        if (kind == GORDER_UA_CH1_UNSAT) {      // uaorder.rs:1024-1045
            const V3 th1 = ps.to(target, c.p0), th2 = ps.to(target, c.p2);
            const float prod = (th1.x * th2.x + th1.y * th2.y) + th1.z * th2.z;
            const float n1 = v3_norm(th1), n2 = v3_norm(th2);
            float gamma = 0.0f;
            if (!(n1 == 0.0f || n2 == 0.0f)) {
                float cs = prod / (n1 * n2);
                cs = cs < -1.0f ? -1.0f : (cs > 1.0f ? 1.0f : cs);
                gamma = gm_acosf(cs);
            }
            const float ang = 3.14159265358979323846f - (gamma / 2.0f);
            const float sn = gm_sinf_0pi(ang), cs = gm_cosf(ang);
            const V3 ua = ps.unit(v3_cross(th1, th2));                                     // perpendicular to th2
            hu = ps.shift(target, ang == 0.0f ? th2 : v3_rotate_perp(ua, sn, cs, th2));
        } else {                                // CH1 saturated, uaorder.rs:1087-1104 (h1, h2, h3, target)
            target = c.p3;
            const V3 t1 = ps.unit(ps.to(target, c.p0)), t2 = ps.unit(ps.to(target, c.p1)), t3 = ps.unit(ps.to(target, c.p2));
            hu = ps.shift(target, V3{-((t1.x + t2.x) + t3.x), -((t1.y + t2.y) + t3.y), -((t1.z + t2.z) + t3.z)});
        }
        r.v0 = bond_v(target, hu);
        r.b0 = bond_b(hu, r.v0);
    }
    slow = ps.slow || pp.slow();
    return r;
}

// the literal-loop variant, kept out of line: it runs only for carbons more than 1.5 box lengths away
// from a helper
__device__ __noinline__ UaBonds ua_carbon_slow(uint32_t kind, UaCarbon c, UaConsts e, V3 box, bool pbc) {
    PbcLoop pl{box, pbc};
    return ua_carbon(kind, c, e, pl);
}

// sums of an int over the DPP rows of a wave (16 lanes each; row shifts: every lane of the wave must be here): lane 15 of
// a row gets the row's sum
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int ua_dpp_add(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true); }
__device__ __forceinline__ int ua_row_sum(int v) {
    v = ua_dpp_add<0x111>(v); v = ua_dpp_add<0x112>(v); v = ua_dpp_add<0x114>(v); v = ua_dpp_add<0x118>(v);
    return v;
}
// ... and on over the rows (row broadcasts): lane 63 gets the wave's sum
__device__ __forceinline__ int ua_rows_to_wave(int v) {
    v = ua_dpp_add<0x142, 0xa>(v); v = ua_dpp_add<0x143, 0xc>(v);
    return v;
}

// MODE 0: order parameters only; 1: + staged ordermap samples, nothing else (no geometry selection, timewise rows or
// per-molecule normals — the common ordermap run, and a much smaller kernel); 2: every extra; 3: per-frame rows, nothing else
// FAST: GORDER_FLAG_UA_FAST_NORMALISE (ua_carbon_fast; inv_box holds 1 / box edge per frame).  PREFETCH: the next frame's
// atoms are requested before this frame's arithmetic (the fast arithmetic no longer hides the gather's latency by itself;
// the exact path is bound by its arithmetic and keeps its registers).
// The body is shared by k_ua_extras (the exact path) and k_ua_extras_fast.
// (Measured and NOT kept in round 4: the tile's needed atoms staged in LDS per frame by coalesced loads, two buffers, one
// barrier per frame — every test green, 0.36 -> 0.47 ms per 3 000 frames for the exact path and 0.30 -> 0.35 for the fast
// one: a frame is only ~300 instructions per lane, a barrier and seventeen LDS operations per frame cost more than the
// gather they replace.)
// (Also measured and not kept: periodic boundaries as a template parameter, so that the `if (!pbc)` in front of every
// minimum image and wrap disappears and the x / y / z chains interleave — 0.284 -> 0.279 ms for the fast path, the exact
// path with maps 0.42 -> 0.49: twice the kernels for nothing.)
template <bool ACOS_COS, int MODE, bool FAST, bool PREFETCH>
__device__ __forceinline__ void ua_extras_body(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                       const float *__restrict__ box9,
                                                       const uint8_t *__restrict__ aflags,
                                                       const uint32_t *__restrict__ arow,
                                                       const Tile *__restrict__ tiles,
                                                       const gorder::UaItem *__restrict__ items,
                                                       const uint32_t *__restrict__ tile_slots, uint32_t n_tiles,
                                                       const float *__restrict__ inv_box) {
    constexpr bool EXTRAS = MODE != 0, GENERAL = MODE >= 2, FULL = MODE == 2, MAPS_POSSIBLE = MODE == 1 || MODE == 2;
    constexpr uint32_t LS = 3 * kBlock;   // local slots per block (<= 3 hydrogens per carbon)
    __shared__ unsigned long long l_s[2 * LS];
    __shared__ uint32_t l_n[2 * LS];
    __shared__ int l_tw[GENERAL ? 3 * LS : 1];
    __shared__ uint32_t l_twn[GENERAL ? 3 * LS : 1];
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    // Workgroups go to the 8 XCDs round-robin; the tiles of one molecule group share their atoms, so each XCD takes a
    // contiguous range of (chunk, tile) pairs — a group's tiles then run next to each other behind ONE L2.
    // (The host pads the grid to a multiple of 8.)
    const uint32_t per_xcd = gridDim.x / 8u, work = (blockIdx.x % 8u) * per_xcd + blockIdx.x / 8u;
    const uint32_t tile_id = work % n_tiles, chunk = work / n_tiles;
    if ((size_t)chunk * a_in.frames_per_chunk >= a_in.n_frames - a_in.frame0) return;    // padding workgroup
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const bool active = tid < t.n_items;
    gorder::UaItem it{};
    if (active) it = items[t.item0 + tid];
    const uint32_t kind = it.kind;
    const int nh = kind == GORDER_UA_CH3 ? 3 : (kind == GORDER_UA_CH2 ? 2 : 1);
    const uint32_t gslot0 = active ? tile_slots[t.slot0 + it.lslot0] : 0;
    const uint32_t f_begin = a.frame0 + chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    const size_t fstride = (size_t)a.n_atoms * 3u;
    // Per-frame rows: when the lanes of every DPP ROW (16 lanes) of the tile hold carbons of ONE kind and slot (the tiles are
    // cut from groups of 64 molecules ordered so that a wave is 4 slots x 16 molecules: the normal case) a frame's sums per
    // slot are sums over a row — DPP adds, then the row's last lane sends them to the rows —: no LDS atomic per hydrogen
    // (two to four, served one lane per cycle) and no barriers per frame.  Otherwise the LDS partials of extras_add /
    // extras_flush_tw.
    __shared__ uint32_t l_mixed;
    if (GENERAL && tid == 0) l_mixed = 0u;
    if (GENERAL)
        for (uint32_t k = tid; k < 3 * LS; k += kBlock) { l_tw[k] = 0; l_twn[k] = 0; }
    for (uint32_t k = tid; k < 2 * LS; k += kBlock) { l_s[k] = 0; l_n[k] = 0; }
    __syncthreads();
    uint32_t row_slot0 = 0;      // the row's first lane: its first slot and its number of hydrogens (active lanes come first)
    int row_nh = 0;
    if (GENERAL) {
        const uint32_t key = active ? ((uint32_t)it.lslot0 << 8) | kind : 0xffffffffu;
        const int lane0 = (int)(tid & 63u & ~15u);
        const uint32_t first = (uint32_t)__shfl((int)key, lane0, 64);
        row_slot0 = (uint32_t)__shfl((int)gslot0, lane0, 64);
        row_nh = first == 0xffffffffu ? 0 : __shfl(nh, lane0, 64);
        if (!__all(!active || key == first)) l_mixed = 1u;
    }
    __syncthreads();
    const bool tw_waves = GENERAL && e.tw && l_mixed == 0u;                     // (uniform over the workgroup)
    const int nh_wave = GENERAL ? (__any(row_nh > 2) ? 3 : (__any(row_nh > 1) ? 2 : 1)) : 0;      // (uniform over the wave)
    // all four rows of the wave hold the same slot (a slot per wave: the lane order of per-frame rows alone): one lane of the
    // wave sends the sums instead of one per row
    const bool wave_one_slot = GENERAL && __all(row_nh == 0 || (row_slot0 == (uint32_t)__builtin_amdgcn_readfirstlane((int)row_slot0) &&
                                                                  row_nh == __builtin_amdgcn_readfirstlane(row_nh)));
    long long s_tot[3] = {0, 0, 0}, s_up[3] = {0, 0, 0};
    uint32_t n_tot[3] = {0, 0, 0}, n_up[3] = {0, 0, 0};
    int bad = 0;
    const bool pbc = a.pbc != 0;
    const UaConsts uc{e.sin_tet, e.cos_tet, e.sin_ch3, e.cos_ch3, e.sin_half, e.cos_half};
    const float axis_x = e.axis == 0 ? 1.0f : 0.0f, axis_y = e.axis == 1 ? 1.0f : 0.0f, axis_z = e.axis == 2 ? 1.0f : 0.0f;
    unsigned long long *rec_row = nullptr;
    uint32_t rec_n = 0;
    const size_t rec_plane = (size_t)kBlock * e.rec_stride;      // hydrogen k of the same lanes: k planes further
    if (MAPS_POSSIBLE && (!GENERAL || e.map_rec) && active) {
        const uint32_t run = e.item_run[t.item0 + tid], tid0 = run >> 16;
        rec_n = run & 0xffffu;
        rec_row = e.map_rec + ((size_t)tile_id * 3u * kBlock + tid0) * e.rec_stride + (tid - tid0);
    }
    const float *src[4];
#pragma unroll
    for (int q = 0; q < 4; q++) src[q] = xyz + ((size_t)t.atom0 + (active ? it.l[q] : 0u)) * 3u;
    auto fetch = [&](uint32_t f) {
        UaCarbon c;
        const size_t o = (size_t)f * fstride;
        c.p0 = {src[0][o], src[0][o + 1], src[0][o + 2]};
        c.p1 = {src[1][o], src[1][o + 1], src[1][o + 2]};
        c.p2 = {src[2][o], src[2][o + 1], src[2][o + 2]};
        c.p3 = {src[3][o], src[3][o + 1], src[3][o + 2]};
        return c;
    };
    // PREFETCH: the next frame's four atoms are asked for before this frame's arithmetic, by EVERY lane (idle lanes point
    // at the tile's first atom) and without a branch (the last frame asks for itself again): inside `if (active)` or
    // behind `f + 1 < f_end` the loaded registers are copied into the loop-carried ones at the end of the conditional
    // block, and the wait for the loads lands right behind them — a prefetch that hides nothing.
    // The frame's box lengths (and their reciprocals): scalar loads at the top of the frame.
    auto fetch_box = [&](uint32_t f, V3 &bx, V3 &inv) {
        bx = {1.0f, 1.0f, 1.0f};
        inv = {1.0f, 1.0f, 1.0f};
        if (pbc) {
            if (FAST) {         // (k_inv_box's record of the frame: one scalar load)
                const float4 *ib = reinterpret_cast<const float4 *>(inv_box) + 2 * (size_t)f;
                const float4 lo = ib[0], hi = ib[1];
                bx = {lo.x, lo.y, lo.z};
                inv = {hi.x, hi.y, hi.z};
            } else {
                const float *b = a.box9 + 9 * (size_t)f;
                bx = {b[0], b[4], b[8]};
            }
        }
    };
    auto fetch_flag = [&](uint32_t f) { return a.aflags[(size_t)a.arow[f] * a.n_mol_total + (active ? it.mol : 0u)]; };
    UaCarbon c_next{};
    uint8_t lf_next = 0;
    if (PREFETCH && f_begin < f_end) {
        c_next = fetch(f_begin);
        if (a.leaflets) lf_next = fetch_flag(f_begin);
    }
    for (uint32_t f = f_begin; f < f_end; f++) {
        int tw_s[3] = {0, 0, 0}, tw_sl[3] = {0, 0, 0}, tw_n[3] = {0, 0, 0};      // tw_waves: this lane's ticks, lower-leaflet ticks, counts (all | lower << 16)
        UaCarbon c_now{};
        V3 bx3{1.0f, 1.0f, 1.0f}, inv3{1.0f, 1.0f, 1.0f};
        if (PREFETCH) {
            c_now = c_next;
            const uint32_t fn = f + 1 < f_end ? f + 1 : f;
            c_next = fetch(fn);
            fetch_box(f, bx3, inv3);        // (scalar loads; a frame ahead they bought nothing and cost six scalar registers)
        } else {
            fetch_box(f, bx3, inv3);
        }
        // (the molecule's leaflet flag: a frame ahead with the atoms — its load is waited for where the flag is first
        // looked at, which the compiler puts right behind the load —, or up here)
        uint8_t lf_raw = 0;
        if (PREFETCH) {
            lf_raw = lf_next;
            if (a.leaflets) lf_next = fetch_flag(f + 1 < f_end ? f + 1 : f);
        } else if (a.leaflets) {
            lf_raw = fetch_flag(f);
        }
        if (active) {
            const UaCarbon c = PREFETCH ? c_now : fetch(f);
            // the atoms are checked in index order (uaorder.rs:400-437 via get_position of each helper); the smallest key wins
            // (one test for the four first — two unordered compares —, the chain only where it fires)
            if (__builtin_expect((int)__builtin_isunordered(c.p0.x, c.p1.x) | (int)__builtin_isunordered(c.p2.x, c.p3.x), 0)) {
                if (c.p0.x != c.p0.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot0, 1, it.mol, 0);
                else if (c.p1.x != c.p1.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot0, 1, it.mol, 1);
                else if (c.p2.x != c.p2.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot0, 1, it.mol, 2);
                else if (c.p3.x != c.p3.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, gslot0, 1, it.mol, 3);
            }
            bool slow = false;
            UaBonds ub;
            if (FAST) {
                ub = ua_carbon_fast<MAPS_POSSIBLE>(kind, c, uc, bx3, inv3, pbc, e.maps != 0 || e.geom_kind != 0, slow);
            } else if (kind == GORDER_UA_CH2 || kind == GORDER_UA_CH3) {       // the two common kinds: paired arithmetic
                ub = ua_carbon_pairs(kind, c, uc, bx3, pbc, slow);
            } else {
                PbcStep ps{bx3, pbc};
                ub = ua_carbon(kind, c, uc, ps);
                slow = ps.slow;
            }
            if (__builtin_expect(slow, 0)) {
                ub = ua_carbon_slow(kind, c, uc, bx3, pbc);
                bad |= ub.bad;
            }
            int leaflet = -1;
            if (a.leaflets) leaflet = lf_raw ? 1 : 0;
            float nrx = a.nx, nry = a.ny, nrz = a.nz, nr2 = a.n2, nr2sq = a.n2sq;
            if (FULL && e.dyn) {   // fetched for every molecule, before the geometry test (uaorder.rs:412-413)
                const float4 n = e.dyn[(size_t)f * a.n_mol_total + it.mol];
                if (n.w < 3.0f) raise_error(a.err, GORDER_ERR_DYNAMIC_NORMAL, f, kStageTypes, gslot0, 1, it.mol, (uint32_t)n.w);
                nrx = n.x; nry = n.y; nrz = n.z;
                nr2sq = (n.x * n.x + n.y * n.y) + n.z * n.z;
                nr2 = __builtin_sqrtf(nr2sq);
            }
            unsigned long long recs[3] = {kMapNoSample, kMapNoSample, kMapNoSample};
            auto sample = [&](const int k, const V3 v, const V3 b) {
                if (k >= nh) return;
                float sch;
                const float s2 = (v.x * v.x + v.y * v.y) + v.z * v.z;
                if (!ACOS_COS && e.axis >= 0 && !(FULL && e.dyn) && s2 >= 0x1p-40f && s2 <= 0x1p+40f) {
                    // static normal along an axis, |v|^2 in the guarded range: the squared cosine by the division core,
                    // no clamp (gm_sch_axis has the argument); anything else takes the general routine
                    // the axis component by a unit vector in scalar registers — x * 1 + y * 0 + z * 0, exact for the finite
                    // components this branch holds (|v|^2 <= 2^40), the sign of a zero aside, which the square drops —:
                    // as selects the two lane masks `axis == 0`, `axis == 1` lived in spilled scalar registers and were
                    // read back (v_readlane) for every hydrogen
                    const float prod = __builtin_fmaf(v.z, axis_z, __builtin_fmaf(v.y, axis_y, v.x * axis_x));
                    sch = (1.5f * gm_div_core(prod * prod, s2)) - 0.5f;
                } else {
                    sch = gm_calc_sch<ACOS_COS>(v.x, v.y, v.z, nrx, nry, nrz, nr2, nr2sq);
                }
                const int tick = gm_tick(sch);
                if (FULL) {
                    const float box[3] = {bx3.x, bx3.y, bx3.z};
                    if (e.geom_kind && !geom_inside(e, e.shapes + 8 * (size_t)f, b.x, b.y, b.z, box, pbc, bad)) return;
                }
                s_tot[k] += tick;
                n_tot[k] += 1;
                if (leaflet == 0) { s_up[k] += tick; n_up[k] += 1; }
                if (tw_waves) {
                    tw_s[k] = tick;
                    tw_sl[k] = leaflet == 1 ? tick : 0;
                    tw_n[k] = 1 | (leaflet == 1 ? 1 << 16 : 0);
                }
                if (EXTRAS)
                    extras_add<!GENERAL, !MAPS_POSSIBLE>(a, e, gslot0 + (uint32_t)k, it.lslot0 + (uint32_t)k, tick, b.x, b.y, b.z, leaflet, l_tw,
                                                         l_twn, LS, (MAPS_POSSIBLE && (!GENERAL || e.map_rec)) ? &recs[k] : nullptr, tw_waves,
                                                         FAST && !slow);
            };
            sample(0, ub.v0, ub.b0);
            sample(1, ub.v1, ub.b1);
            sample(2, ub.v2, ub.b2);
            if (MAPS_POSSIBLE && (!GENERAL || e.map_rec)) {   // one word per hydrogen this carbon has, into its run's piece
                unsigned long long *row = rec_row + (size_t)(f - e.rec_frame0) * rec_n;
                row[0] = recs[0];
                if (nh > 1) row[rec_plane] = recs[1];
                if (nh > 2) row[2u * rec_plane] = recs[2];
            }
        }
        if (tw_waves) {             // every lane of the wave is here
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (k >= nh_wave) break;                            // (uniform)
                int s_all = ua_row_sum(tw_s[k]), s_low = ua_row_sum(tw_sl[k]), n = ua_row_sum(tw_n[k]);
                if (wave_one_slot) { s_all = ua_rows_to_wave(s_all); s_low = ua_rows_to_wave(s_low); n = ua_rows_to_wave(n); }
                const uint32_t n_all = (uint32_t)n & 0xffffu, n_low = (uint32_t)n >> 16;
                const bool sender = wave_one_slot ? (tid & 63u) == 63u : (tid & 15u) == 15u;
                // (a wave that ends inside the tile: its last rows are idle, the wave's slot is that of its first lane)
                const int send_nh = wave_one_slot ? __builtin_amdgcn_readfirstlane(row_nh) : row_nh;
                const uint32_t send_slot0 = wave_one_slot ? (uint32_t)__builtin_amdgcn_readfirstlane((int)row_slot0) : row_slot0;
                if (sender && k < send_nh && n_all) {
                    const size_t row = ((size_t)e.tw_row0 + f) * 3u * a.n_acc;
                    const uint32_t slot = send_slot0 + (uint32_t)k;
                    atomicAdd(&e.tw_sums[row + slot], (unsigned long long)(long long)s_all);
                    atomicAdd(&e.tw_cnts[row + slot], (unsigned long long)n_all);
                    if (a.leaflets) {
                        const uint32_t n_up_w = n_all - n_low;
                        if (n_up_w) {
                            atomicAdd(&e.tw_sums[row + (size_t)a.n_acc + slot], (unsigned long long)(long long)(s_all - s_low));
                            atomicAdd(&e.tw_cnts[row + (size_t)a.n_acc + slot], (unsigned long long)n_up_w);
                        }
                        // (the lower leaflet's row: total - upper, gorder_hip_timewise)
                    }
                }
            }
        } else if (GENERAL && e.tw) {
            __syncthreads();
            extras_flush_tw(a, e, tile_slots + t.slot0, t.n_slots, f, l_tw, l_twn, LS);
            __syncthreads();
        }
    }
    if (bad) raise_box_range(a.err, f_begin);
    if (active) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (!n_tot[k]) continue;
            atomicAdd(&l_s[it.lslot0 + k], (unsigned long long)s_tot[k]);
            atomicAdd(&l_n[it.lslot0 + k], n_tot[k]);
            if (n_up[k]) {
                atomicAdd(&l_s[LS + it.lslot0 + k], (unsigned long long)s_up[k]);
                atomicAdd(&l_n[LS + it.lslot0 + k], n_up[k]);
            }
        }
    }
    __syncthreads();
    unsigned long long *accp = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
    for (uint32_t ls = tid; ls < t.n_slots; ls += kBlock) {
        if (!l_n[ls]) continue;
        const uint32_t slot = tile_slots[t.slot0 + ls];
        atomicAdd(&accp[slot], l_s[ls]);
        atomicAdd(&accp[2u * a.n_acc + slot], (unsigned long long)l_n[ls]);
        if (l_n[LS + ls]) {
            atomicAdd(&accp[a.n_acc + slot], l_s[LS + ls]);
            atomicAdd(&accp[3u * a.n_acc + slot], (unsigned long long)l_n[LS + ls]);
        }
    }
}

#ifndef GORDER_UA_PREFETCH
#define GORDER_UA_PREFETCH 0        // the exact kernel is bound by VALU issue: a frame ahead buys nothing there (measured, round 4)
#endif
#define GORDER_UA_KERNEL_ARGS                                                                                              \
    FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz, const float *__restrict__ box9,                               \
        const uint8_t *__restrict__ aflags, const uint32_t *__restrict__ arow, const Tile *__restrict__ tiles,                \
        const gorder::UaItem *__restrict__ items, const uint32_t *__restrict__ tile_slots, uint32_t n_tiles,                  \
        const float *__restrict__ inv_box
template <bool ACOS_COS, int MODE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_ua_extras(GORDER_UA_KERNEL_ARGS) {
    ua_extras_body<ACOS_COS, MODE, false, GORDER_UA_PREFETCH != 0>(a_in, e, xyz, box9, aflags, arow, tiles, items, tile_slots, n_tiles, inv_box);
}
// GORDER_FLAG_UA_FAST_NORMALISE (the default cosine only)
#ifndef GORDER_UA_FAST_WAVES
#define GORDER_UA_FAST_WAVES 4      // (3, 5 and 6 waves per SIMD measured in round 4: see DESIGN)
#endif
template <int MODE>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(GORDER_UA_FAST_WAVES, GORDER_UA_FAST_WAVES))) void k_ua_extras_fast(GORDER_UA_KERNEL_ARGS) {
    ua_extras_body<false, MODE, true, true>(a_in, e, xyz, box9, aflags, arow, tiles, items, tile_slots, n_tiles, inv_box);
}
#undef GORDER_UA_KERNEL_ARGS

// ---- ordermaps, second step ------------------------------------------------------------------------
// The sample kernels stage every sample as five bytes (tick, plane-tile: ExtraArgs::map_rec), run by run; here a block
// owns ONE accumulator slot for a range of frames: it reads the slot's runs (gorder::MapRun, contiguous pieces),
// adds the samples into a packed map held in LDS (ds_add_u64) and flushes the tiles it touched
// into the global packed map with one atomic each.  Scattered global atomics run at ~24 G/s on this chip
// whatever one does (tools/microbench/atomic_scatter.hip); this way their number drops from one per sample to
// at most one per (slot, chunk, tile).
constexpr uint32_t kMapRunBatch = 1024;          // runs of a slot whose lanes k_map_accumulate lays out at a time
__global__ __launch_bounds__(1024) void k_map_accumulate(const unsigned long long *__restrict__ rec,
                                                         const gorder::MapRun *__restrict__ runs,
                                                         const uint32_t *__restrict__ run_begin, uint32_t n_slots,
                                                         uint32_t rec_frames, uint32_t rec_stride,
                                                         uint32_t frames_per_chunk, uint32_t k_max,
                                                         uint32_t n_words /* planes * tiles */, uint32_t n_tiles_map,
                                                         unsigned long long *__restrict__ map_packed, uint32_t n_acc) {
    extern __shared__ unsigned long long l_map[];
    typedef unsigned long long ull2 __attribute__((ext_vector_type(2)));
    const uint32_t slot = blockIdx.x % n_slots, chunk = blockIdx.x / n_slots;
    const uint32_t r0 = run_begin[slot], r1 = run_begin[slot + 1];
    if (r0 == r1) return;                               // no samples of this kind (bond / united atom) in the slot
    const uint32_t f0 = chunk * frames_per_chunk, f1 = min(rec_frames, f0 + frames_per_chunk);
    for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) l_map[w] = 0ull;
    __syncthreads();
    auto add = [&](unsigned long long v) {
        if (v != kMapNoSample) atomicAdd(&l_map[(uint32_t)(v >> 32)], kMapOne + (unsigned long long)(long long)(int)(uint32_t)v);
    };
    if (k_max == 1u) {
        // Bond tiles.  A lane takes ONE molecule of the slot — lane j of a run — and walks its samples through the frame
        // blocks of its share of the chunk, in time order (the addresses are base + block * stride: no run table in the
        // loop, the next block's loads go out before this one's samples are added); samples that follow each other into
        // the same tile — a molecule moves less than a tile between most frames — are added up in registers and cost
        // ONE ds_add_u64.
        __shared__ uint32_t l_pref[kMapRunBatch + 1];
        const uint32_t fb0 = f0 / kRecFrames, fb1 = (f1 + kRecFrames - 1) / kRecFrames;
        for (uint32_t rb = r0; rb < r1; rb += kMapRunBatch) {
            const uint32_t nr = min(kMapRunBatch, r1 - rb);
            for (uint32_t i = threadIdx.x; i < nr; i += blockDim.x) l_pref[i + 1u] = runs[rb + i].n;
            if (threadIdx.x == 0) l_pref[0] = 0;
            __syncthreads();
            if (threadIdx.x < 64u) {            // inclusive scan by one wave
                uint32_t carry = 0;
                for (uint32_t i0 = 1; i0 <= nr; i0 += 64u) {
                    const uint32_t i = i0 + threadIdx.x, v = i <= nr ? l_pref[i] : 0u;
                    uint32_t incl = v;
                    for (uint32_t off = 1; off < 64u; off <<= 1) {
                        const uint32_t u = __shfl_up(incl, off, 64);
                        if (threadIdx.x >= off) incl += u;
                    }
                    if (i <= nr) l_pref[i] = carry + incl;
                    carry += __shfl(incl, 63, 64);
                }
            }
            __syncthreads();
            const uint32_t total = l_pref[nr];
            // fewer molecules than threads: the frame blocks of the chunk are shared out among `parts` lanes per molecule
            const uint32_t parts = total ? max(1u, min(blockDim.x / total, fb1 - fb0)) : 1u;
            const uint32_t fb_per = (fb1 - fb0 + parts - 1u) / parts;
            for (uint32_t lane_item = threadIdx.x; lane_item < total * parts; lane_item += blockDim.x) {
                const uint32_t item = lane_item % total, part = lane_item / total;
                const uint32_t fb_lo = fb0 + part * fb_per, fb_hi = min(fb1, fb_lo + fb_per);
                uint32_t lo = 0, hi = nr;                      // the run with l_pref[lo] <= item < l_pref[lo + 1]
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (l_pref[mid] <= item) lo = mid; else hi = mid;
                }
                const gorder::MapRun run = runs[rb + lo];
                const unsigned long long *base = rec + (size_t)run.tile * kBlock * rec_stride + kRecFrames * (run.tid0 + (item - l_pref[lo]));
                uint32_t cur = 0;
                unsigned long long sum = 0ull;
                ull2 wa, wb, na, nb;
                auto fetch = [&](uint32_t fb, ull2 &a2, ull2 &b2) {
                    const ull2 *p = reinterpret_cast<const ull2 *>(base + (size_t)fb * (kRecFrames * kBlock));
                    a2 = __builtin_nontemporal_load(p);
                    b2 = __builtin_nontemporal_load(p + 1);
                };
                if (fb_lo < fb_hi) fetch(fb_lo, na, nb);
                for (uint32_t fb = fb_lo; fb < fb_hi; fb++) {
                    wa = na; wb = nb;
                    if (fb + 1u < fb_hi) fetch(fb + 1u, na, nb);
                    const unsigned long long w[4] = {wa.x, wa.y, wb.x, wb.y};
#pragma unroll
                    for (uint32_t m = 0; m < 4u; m++) {
                        if (w[m] == kMapNoSample) continue;
                        const uint32_t pt = (uint32_t)(w[m] >> 32);
                        if (sum != 0ull && pt != cur) { atomicAdd(&l_map[cur], sum); sum = 0ull; }
                        cur = pt;
                        sum += kMapOne + (unsigned long long)(long long)(int)(uint32_t)w[m];
                    }
                }
                if (sum != 0ull) atomicAdd(&l_map[cur], sum);
            }
            __syncthreads();
        }
    } else {
        // united-atom tiles: few long runs — the whole block strides over each run's contiguous piece of the frames
        // [f0, f1) —, or many short ones: a wave per run at a time
        const uint32_t n_waves = blockDim.x >> 6;
        const bool per_wave = r1 - r0 >= n_waves;
        const uint32_t r_first = r0 + (per_wave ? threadIdx.x >> 6 : 0u), r_step = per_wave ? n_waves : 1u;
        const uint32_t i_first = per_wave ? threadIdx.x & 63u : threadIdx.x, i_step = per_wave ? 64u : blockDim.x;
        for (uint32_t r = r_first; r < r1; r += r_step) {
            const gorder::MapRun run = runs[r];
            const uint32_t total = (f1 - f0) * run.n;
            const unsigned long long *piece = rec + (((size_t)run.tile * k_max + run.k) * kBlock + run.tid0) * rec_stride + (size_t)f0 * run.n;
            // (the piece starts at a multiple of 16 words: pairs — 16-byte loads —, then the last chunk's odd word)
            for (uint32_t i = i_first; i < total / 2u; i += i_step) {
                const ull2 w = __builtin_nontemporal_load(reinterpret_cast<const ull2 *>(piece) + i);
                add(w.x);
                add(w.y);
            }
            if ((total & 1u) && i_first == 0u) add(piece[total - 1u]);
        }
    }
    __syncthreads();
    for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) {
        const unsigned long long v = l_map[w];
        if (!v) continue;
        const uint32_t plane = w / n_tiles_map, t = w % n_tiles_map;
        atomicAdd(&map_packed[((size_t)plane * n_acc + slot) * n_tiles_map + t], v);
    }
}

}  // namespace
