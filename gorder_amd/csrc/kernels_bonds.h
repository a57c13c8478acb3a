// kernels_bonds.h — K1 k_bonds_tiled (+ gather / direct variants): BondType::analyze_frame, bond.rs:396-446.
// Part of the single translation unit gorder_hip.hip (included there, in this order: common, bonds, extras,
// leaflets, normals); device code for gfx950 only.
#pragma once

namespace {

// groan_rs GridMap::get_mut_at: nearest tile centre, None outside (oracle: gridmap_index)
// `core`: the bin width lies in [2^-40, 2^40] (checked once on the host), so the quotient can come from the Newton
// core of the IEEE division (gm_div_core): a numerator small or large enough for v_div_scale to matter gives a tile
// index of 0 or none either way.  The reciprocal refinement is loop-invariant (one bin width for all samples).
__device__ __forceinline__ int grid_index(float x, float lo, float bin, uint32_t n, bool core) {
    const float k = __builtin_roundf(core ? gm_div_core(x - lo, bin) : (x - lo) / bin);
    if (!(k >= 0.0f) || !(k < (float)n)) return -1;
    return (int)k;
}
// GORDER_FLAG_UA_FAST_NORMALISE: the tile by one fma and a floor — floor((x - lo) * (1 / bin) + 0.5) — instead of the
// rounded IEEE quotient: a sample within an ulp or two of the line between two tiles may land on the other side
// (oracle: gridmap_index_fast; tools/ua_fast_fidelity.py counts them).
__device__ __forceinline__ int grid_index_fast(float x, float lo, float inv_bin, uint32_t n) {
    const float k = __builtin_floorf(__builtin_fmaf(x - lo, inv_bin, 0.5f));
    if (!(k >= 0.0f) || !(k < (float)n)) return -1;
    return (int)k;
}
// Ordermap samples out of the tiled kernel (k_bonds_tiled_maps, kernels_extras.h): the grid of the map and this
// thread's words of the stage's kRecFrames frames.
constexpr uint32_t kRecFrames = 4;               // frames per block of the bond tiles' staging layout = frames of a stage
struct TiledMapOut {
    uint32_t plane;                  // 0 xy, 1 xz, 2 yz -> (z, y)
    float x0, y0, binx, biny;
    uint32_t nx, ny;
    int bin_core;
    unsigned long long *words;
    unsigned long long *ticks;       // OUT = 3: the (side, tick) words next to the ordermap words
};
// what k_bonds_tiled_tw wants of a sample: its side and its tick
__device__ __forceinline__ unsigned long long tick_sample_word(int tick, bool lower) {
    return ((unsigned long long)(lower ? 1u : 0u) << 32) | (unsigned long long)(uint32_t)tick;
}
// the staged word of one sample: (plane-of-leaflet * tiles + tile) << 32 | tick, or kMapNoSample outside the map
__device__ __forceinline__ unsigned long long map_sample_word(const TiledMapOut &mo, float px, float py, float pz, int tick, bool lower) {
    float x, y;
    if (mo.plane == 0) { x = px; y = py; }
    else if (mo.plane == 1) { x = px; y = pz; }
    else { x = pz; y = py; }
    const int ix = grid_index(x, mo.x0, mo.binx, mo.nx, mo.bin_core != 0), iy = grid_index(y, mo.y0, mo.biny, mo.ny, mo.bin_core != 0);
    if (ix < 0 || iy < 0) return kMapNoSample;
    const uint32_t nt = mo.nx * mo.ny, t = (uint32_t)ix * mo.ny + (uint32_t)iy;
    return ((unsigned long long)((lower ? nt : 0u) + t) << 32) | (unsigned long long)(uint32_t)tick;
}

// ---- one bond sample (bond.rs:407-443) -----------------------------------------------------
constexpr uint32_t kNoNan = 0xffffffffu;   // "this sample has met no undefined position yet"
struct SampleAcc {
    long long s_tot = 0, s_up = 0;
    uint32_t n_tot = 0, n_up = 0;
};

// returns true when S came out NaN (undefined position or a non-finite coordinate)
template <bool ACOS_COS>
__device__ __forceinline__ bool bond_sample(const FrameArgs &a, uint32_t f, float p1x, float p1y, float p1z,
                                            float p2x, float p2y, float p2z, uint32_t mol, SampleAcc &acc,
                                            int &bad, const TiledMapOut *mo = nullptr, unsigned long long *word = nullptr,
                                            unsigned long long *tick_word = nullptr) {
    float vx = p2x - p1x, vy = p2y - p1y, vz = p2z - p1z;
    if (a.pbc) {
        const float *b = a.box9 + 9 * (size_t)f;
        const float bx = b[0], by = b[4], bz = b[8];
        bool slow = false;   // one select-only step per dimension; the literal loops only when needed
        const float rx = gm_min_image_step(vx, bx, slow);
        const float ry = gm_min_image_step(vy, by, slow);
        const float rz = gm_min_image_step(vz, bz, slow);
        if (__builtin_expect(slow, 0)) {
            vx = gm_min_image_loop(vx, bx, bad);
            vy = gm_min_image_loop(vy, by, bad);
            vz = gm_min_image_loop(vz, bz, bad);
        } else {
            vx = rx; vy = ry; vz = rz;
        }
    }
    const float sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
    const long long tick = gm_tick(sch);
    acc.s_tot += tick;
    acc.n_tot += 1;
    bool lower = false;
    if (a.leaflets) {
        const uint8_t fl = a.aflags[(size_t)a.arow[f] * a.n_mol_total + mol];
        lower = fl != 0;
        if (fl == 0) {   // Leaflet::Upper = 0 (lib.rs:416-422)
            acc.s_up += tick;
            acc.n_up += 1;
        }
    }
    if (mo) {
        *word = mo->nx == 0u ? tick_sample_word((int)tick, lower)        // (k_bonds_tiled_tw)
                             : map_sample_word(*mo, p1x + vx / 2.0f, p1y + vy / 2.0f, p1z + vz / 2.0f, (int)tick, lower);   // bond position = p1 + v / 2 (bond.rs:422)
        if (tick_word) *tick_word = tick_sample_word((int)tick, lower);
    }
    return sch != sch;
}

// ---- K1: tiled bonds ----------------------------------------------------------------------
// grid.x = n_tiles * n_chunks; block = 256 = 4 waves; dynamic LDS = G * lw floats.
// Each block owns one tile (<= 256 samples, one contiguous atom window) for frames_per_chunk frames.
// Per stage the window of G frames goes HBM -> registers -> LDS (256/G threads per frame, 16 B per
// lane, fully coalesced, every byte read once) and every thread evaluates its sample for the G
// frames.  The loads of stage s+1 are issued BEFORE the arithmetic of stage s (NPF float4 registers
// per thread), so each resident block keeps a whole stage of HBM traffic in flight while it computes.
//
// The 16-byte loads start at the window's first float rounded DOWN to 16 B and end at its last float
// rounded UP to 16 B.  xyz is 16-byte aligned, so the last load of the whole buffer stays inside the
// aligned 16-byte granule that holds the last valid float: it cannot cross into an unmapped page.
#ifndef GORDER_COMPUTE_NF
#define GORDER_COMPUTE_NF 8        // frames evaluated together per thread (capped at G)
#endif
typedef float v4f __attribute__((ext_vector_type(4)));   // native 16-byte vector (SROA-friendly, unlike float4)

template <int G, int NPF, bool ACOS_COS, bool PBC, bool LEAF, int AXIS = -1>
struct TiledStage {
    static constexpr uint32_t TPF = kBlock / G;   // threads that stage one frame

    // issue the loads of my frame slot of the stage that starts at frame f0
    template <bool TAIL>
    static __device__ __forceinline__ void load(const FrameArgs &a, const Tile &t, uint32_t f0, uint32_t f_end,
                                                uint32_t sk, uint32_t si, v4f (&pre)[NPF]) {
        const uint32_t f = f0 + sk;
        if (TAIL && f >= f_end) return;
        const size_t base = ((size_t)f * a.n_atoms + t.atom0) * 3u;
        const uint32_t n4 = ((uint32_t)(base & 3u) + 3u * t.n_window + 3u) >> 2;
        const v4f *src = reinterpret_cast<const v4f *>(a.xyz + (base & ~(size_t)3));
#pragma unroll
        for (int j = 0; j < NPF; j++) {   // unconditional (index clamped): keeps pre[] in registers
            const uint32_t i = si + (uint32_t)j * TPF;
            pre[j] = __builtin_nontemporal_load(src + (i < n4 ? i : n4 - 1u));   // streamed once: nt
        }
    }
    // registers (and, for windows wider than NPF * TPF float4, late loads) -> LDS
    // (FINITE: also says whether every float this thread staged is finite — MOM's stand-in for the reference's check of the
    // membrane atoms' coordinates; the 16-byte granules may hold a neighbour's float or two: a NaN there only sends the
    // frame to the exact kernel)
    template <bool TAIL, bool FINITE = false>
    static __device__ __forceinline__ bool store(const FrameArgs &a, const Tile &t, uint32_t f0, uint32_t f_end,
                                                 uint32_t sk, uint32_t si, const v4f (&pre)[NPF], float *lds,
                                                 uint32_t lw) {
        const uint32_t f = f0 + sk;
        if (TAIL && f >= f_end) return true;
        const size_t base = ((size_t)f * a.n_atoms + t.atom0) * 3u;
        const uint32_t n4 = ((uint32_t)(base & 3u) + 3u * t.n_window + 3u) >> 2;
        const v4f *src = reinterpret_cast<const v4f *>(a.xyz + (base & ~(size_t)3));
        v4f *dst = reinterpret_cast<v4f *>(lds + (size_t)sk * lw);
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f poison = {0.0f, 0.0f};                  // stays 0 while everything is finite: 0 * x is 0, or NaN for an infinity or a NaN
        const v2f zero2 = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const uint32_t i = si + (uint32_t)j * TPF;
            if (i < n4) dst[i] = pre[j];
            if (FINITE) {
                poison = __builtin_elementwise_fma(pre[j].xy, zero2, poison);
                poison = __builtin_elementwise_fma(pre[j].zw, zero2, poison);
            }
        }
        for (uint32_t i = si + (uint32_t)NPF * TPF; i < n4; i += TPF) {
            const v4f v = __builtin_nontemporal_load(src + i);
            dst[i] = v;
            if (FINITE) {
                poison = __builtin_elementwise_fma(v.xy, zero2, poison);
                poison = __builtin_elementwise_fma(v.zw, zero2, poison);
            }
        }
        return poison.x + poison.y == 0.0f;
    }
    // My sample in each of the G frames of a stage; P[k] = {p1x,p1y,p1z,p2x,p2y,p2z} of frame f0 + k.
    // The common path is straight-line code (selects only) so that the G independent dependency chains
    // interleave; the rare cases (atoms more than 1.5 box lengths apart -> literal minimum-image loops;
    // NaN result -> which atom is undefined?) are collected in a bit mask and handled after the stage.
    template <int NF, int OUT = 0>       // OUT: 0 the accumulators only, 1 + the ordermap words, 2 + (side, tick) words, 3 + both
    static __device__ __forceinline__ void compute_core(const FrameArgs &a, const Tile &t, const Item &it,
                                                        uint32_t f0, const float (&P)[NF][6], SampleAcc &acc,
                                                        int &bad, uint32_t &nan_which, uint32_t &nan_frame,
                                                        const TiledMapOut *mo = nullptr, uint32_t k0 = 0) {
        constexpr bool MAPS = OUT != 0;
        int tick[NF];
        uint8_t fl[NF];
        float bx[NF], by[NF], bz[NF];
        uint32_t rare = 0;
        // MAPS: the sample's word (bond position = p1 + v / 2, bond.rs:422, v the SIGNED minimum-image vector) goes to
        // this thread's word of frame k0 + k of the stage as soon as its tick is known
        auto stage_word = [&](int k, float vx, float vy, float vz) {
            mo->words[k0 + (uint32_t)k] = OUT == 2
                ? tick_sample_word(tick[k], LEAF && fl[k] != 0)
                : map_sample_word(*mo, P[k][0] + vx / 2.0f, P[k][1] + vy / 2.0f, P[k][2] + vz / 2.0f, tick[k], LEAF && fl[k] != 0);
            if (OUT == 3) mo->ticks[k0 + (uint32_t)k] = tick_sample_word(tick[k], LEAF && fl[k] != 0);
        };
        // uniform per-frame inputs of the whole stage first (scalar loads, issued back to back)
#pragma unroll
        for (int k = 0; k < NF; k++) {
            if (PBC) {
                const float *b = a.box9 + 9 * (size_t)(f0 + k);
                bx[k] = b[0]; by[k] = b[4]; bz[k] = b[8];
            }
            if (LEAF) fl[k] = a.aflags[(size_t)a.arow[f0 + k] * a.n_mol_total + it.mol];
        }
#pragma unroll
        for (int k = 0; k < NF; k++) {
            float vx = P[k][3] - P[k][0], vy = P[k][4] - P[k][1], vz = P[k][5] - P[k][2];
            bool slow = false;
            if (PBC && AXIS >= 0 && !ACOS_COS && !MAPS) {
                // only squares are taken below, so magnitudes are enough: |dx - copysign(L, dx)| = |L - |dx|| bit for
                // bit.  One shift suffices while |dx| <= L (then L - |dx| >= 0 and <= L/2): a negative L - |dx| in any
                // dimension sends the sample to the literal loops (a superset of gm_min_image_step's `slow`).
                const float ax = __builtin_fabsf(vx), ay = __builtin_fabsf(vy), az = __builtin_fabsf(vz);
                const float tx = bx[k] - ax, ty = by[k] - ay, tz = bz[k] - az;
                vx = ax > bx[k] / 2.0f ? tx : ax;
                vy = ay > by[k] / 2.0f ? ty : ay;
                vz = az > bz[k] / 2.0f ? tz : az;
                slow = __builtin_fminf(__builtin_fminf(tx, ty), tz) < 0.0f;
            } else if (PBC) {
                vx = gm_min_image_step(vx, bx[k], slow);
                vy = gm_min_image_step(vy, by[k], slow);
                vz = gm_min_image_step(vz, bz[k], slow);
            }
            if (AXIS >= 0 && !ACOS_COS) {     // the headline case: static normal along an axis
                const float sch = gm_sch_axis<AXIS>(vx, vy, vz, slow);
                rare |= (slow ? 1u : 0u) << k;
                tick[k] = gm_tick_finite(sch);
            } else if (AXIS >= 0) {           // the same normal, the reference's literal acos -> cos round trip
                const float sch = gm_sch_axis_acos<AXIS>(vx, vy, vz, slow);
                rare |= (slow ? 1u : 0u) << k;
                tick[k] = gm_tick_finite(sch);
            } else {
                bool nonfinite = false;
                const float sch = gm_calc_sch<ACOS_COS, AXIS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq, &nonfinite);
                rare |= ((slow || nonfinite || sch != sch) ? 1u : 0u) << k;
                tick[k] = gm_tick(sch);
            }
            if (MAPS) stage_word(k, vx, vy, vz);
        }
        if (__builtin_expect(rare != 0, 0)) {
#pragma unroll
            for (int k = 0; k < NF; k++) {
                if (!((rare >> k) & 1u)) continue;
                float vx = P[k][3] - P[k][0], vy = P[k][4] - P[k][1], vz = P[k][5] - P[k][2];
                if (PBC) {
                    vx = gm_min_image_loop(vx, bx[k], bad);
                    vy = gm_min_image_loop(vy, by[k], bad);
                    vz = gm_min_image_loop(vz, bz[k], bad);
                }
                const float sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
                tick[k] = gm_tick(sch);
                if (MAPS) stage_word(k, vx, vy, vz);
                if (sch != sch && nan_frame == kNoNan) {   // the FIRST undefined position this sample meets (bond.rs:410-416)
                    if (P[k][0] != P[k][0]) { nan_which = 0; nan_frame = f0 + k; }
                    else if (P[k][3] != P[k][3]) { nan_which = 1; nan_frame = f0 + k; }
                }
            }
        }
        int st = 0, su = 0;
        uint32_t nu = 0;
#pragma unroll
        for (int k = 0; k < NF; k++) {
            st += tick[k];
            if (LEAF) {   // Leaflet::Upper = 0 (lib.rs:416-422)
                su += fl[k] == 0 ? tick[k] : 0;
                nu += fl[k] == 0 ? 1u : 0u;
            }
        }
        acc.s_tot += st;
        acc.n_tot += NF;
        acc.s_up += su;
        acc.n_up += nu;
    }
    // LDS-staged variant: pick my two atoms out of the staged windows
    template <int OUT = 0>
    static __device__ __forceinline__ void compute(const FrameArgs &a, const Tile &t, const Item &it, uint32_t f0,
                                                   const float *lds, uint32_t lw, SampleAcc &acc, int &bad,
                                                   uint32_t &nan_which, uint32_t &nan_frame, const TiledMapOut *mo = nullptr) {
        // NF frames at a time: NF independent dependency chains interleave; fewer live registers than all G at once
        // (with the map words two at a time: four chains and their tile arithmetic do not fit the 128 registers)
        constexpr int NF = (OUT == 1 || OUT == 3 || (OUT == 2 && LEAF)) ? 2 : (GORDER_COMPUTE_NF < G ? GORDER_COMPUTE_NF : G);
#pragma unroll
        for (int h = 0; h < G; h += NF) {
            float P[NF][6];
#pragma unroll
            for (int k = 0; k < NF; k++) {
                const uint32_t sh = (uint32_t)((((size_t)(f0 + h + k) * a.n_atoms + t.atom0) * 3u) & 3u);
                const float *w = lds + (size_t)(h + k) * lw + sh;
                P[k][0] = w[3u * it.li]; P[k][1] = w[3u * it.li + 1]; P[k][2] = w[3u * it.li + 2];
                P[k][3] = w[3u * it.lj]; P[k][4] = w[3u * it.lj + 1]; P[k][5] = w[3u * it.lj + 2];
            }
            compute_core<NF, OUT>(a, t, it, f0 + h, P, acc, bad, nan_which, nan_frame, mo, (uint32_t)h);
        }
    }
    // partial last stage: frames f0 .. f_end-1, one at a time (not performance relevant)
    template <int OUT = 0>
    static __device__ __forceinline__ void compute_tail(const FrameArgs &a, const Tile &t, const Item &it,
                                                        uint32_t f0, uint32_t f_end, const float *lds, uint32_t lw,
                                                        SampleAcc &acc, int &bad, uint32_t &nan_which,
                                                        uint32_t &nan_frame, const TiledMapOut *mo = nullptr) {
        if (OUT != 0) {     // (unrolled: the words stay in registers)
#pragma unroll
            for (int k = 0; k < G; k++) {
                const uint32_t f = f0 + (uint32_t)k;
                if (f >= f_end) continue;
                const uint32_t sh = (uint32_t)((((size_t)f * a.n_atoms + t.atom0) * 3u) & 3u);
                const float *w = lds + (size_t)k * lw + sh;
                const float p1x = w[3u * it.li], p1y = w[3u * it.li + 1], p1z = w[3u * it.li + 2];
                const float p2x = w[3u * it.lj], p2y = w[3u * it.lj + 1], p2z = w[3u * it.lj + 2];
                if (bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad, mo, mo->words + k, OUT == 3 ? mo->ticks + k : nullptr) &&
                    nan_frame == kNoNan) {
                    if (p1x != p1x) { nan_which = 0; nan_frame = f; }
                    else if (p2x != p2x) { nan_which = 1; nan_frame = f; }
                }
            }
            return;
        }
#pragma unroll 1
        for (uint32_t f = f0; f < f_end; f++) {
            const uint32_t sh = (uint32_t)((((size_t)f * a.n_atoms + t.atom0) * 3u) & 3u);
            const float *w = lds + (size_t)(f - f0) * lw + sh;
            const float p1x = w[3u * it.li], p1y = w[3u * it.li + 1], p1z = w[3u * it.li + 2];
            const float p2x = w[3u * it.lj], p2y = w[3u * it.lj + 1], p2z = w[3u * it.lj + 2];
            if (bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad) && nan_frame == kNoNan) {
                if (p1x != p1x) { nan_which = 0; nan_frame = f; }
                else if (p2x != p2x) { nan_which = 1; nan_frame = f; }
            }
        }
    }
};

#ifndef GORDER_TILED_MIN_WAVES
#define GORDER_TILED_MIN_WAVES 4   // waves per SIMD the register allocation must allow (8 => <= 64 VGPRs)
#endif
// MOM (one read for global leaflets + order parameters): the wave that stands for frame slot k of the stage sums the
// normal coordinate of the atoms this tile OWNS (FrameArgs::own) out of the window that is in LDS anyway — relative to the
// middle of the box, with its square and the extrema; a NaN coordinate poisons the sums — and leaves them for
// k_spec_resolve: mom[frame][tile].  Five instructions an atom, no sine or cosine.
// a wave reduction by DPP alone: four shifts inside the rows of 16, then the rows' results passed on (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into the upper half): the total ends up in lane 63.  Sums take 0 where the control names no
// lane (bound_ctrl: the shift folds into the add), extrema their own value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mom_dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mom_dpp_self(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// v_min_f32 / v_max_f32 as the hardware has them: `fminf` makes the compiler canonicalise both operands first (a v_max x, x, x
// each — over a hundred extra instructions in the moments), and a NaN does not matter here: the finiteness of the staged
// floats is checked on its own.  The DPP forms write only the lanes whose source lane exists (the others keep their value);
// the s_nop covers the two wait states a DPP read needs behind a VALU write, which inline assembly has to provide itself.
__device__ __forceinline__ float mom_min(float x, float y) { float r; __asm__("v_min_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float mom_max(float x, float y) { float r; __asm__("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
#define GORDER_MOM_MINMAX_DPP(CTRL_TEXT)                                                                    \
    __asm__("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL_TEXT "\n\tv_max_f32_dpp %1, %1, %1 " CTRL_TEXT : "+v"(mn), "+v"(mx))
__device__ __forceinline__ void tiled_moments(const FrameArgs &a, const Tile &t, uint2 own, uint32_t tile_id, uint32_t n_tiles,
                                              uint32_t f, const float *slot, bool finite, uint2 my_head) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sh = (uint32_t)((((size_t)f * a.n_atoms + t.atom0) * 3u) & 3u);
    const float *w = slot + sh + a.mom_dim + 3u * (own.x - t.atom0);
    const uint32_t n_own = own.y - own.x;
    float s = 0.0f, q = 0.0f, mn = 3.0e38f, mx = -3.0e38f;
    // eight atoms a lane and trip, their LDS reads in flight together (one trip for up to 512 owned atoms: the moments sit
    // between a stage's loads and its arithmetic, and what they wait for the stage waits for); plain sums of z and z^2 —
    // k_spec_check adds the tiles' in f64 and allows for the f32 rounding here.  A lane past the range reads the range's
    // last atom again: that leaves the extrema alone, and only the one group of 64 that is not full masks its sums.
    for (uint32_t i0 = 0; i0 < n_own; i0 += 512u) {
        float z[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) z[u] = w[3u * min(i0 + 64u * u + lane, n_own - 1u)];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            if (i0 + 64u * u >= n_own) break;                              // (uniform)
            float d = z[u];
            if (i0 + 64u * u + 64u > n_own) d = i0 + 64u * u + lane < n_own ? d : 0.0f;     // (uniform condition: the partial group)
            s += d;
            q = __builtin_fmaf(d, d, q);
            mn = mom_min(mn, z[u]);
            mx = mom_max(mx, z[u]);
        }
    }
#define GORDER_MOM_STEP(CTRL, MASK, CTRL_TEXT)                                                  \
    s += mom_dpp0<CTRL, MASK>(s); q += mom_dpp0<CTRL, MASK>(q);                                 \
    GORDER_MOM_MINMAX_DPP(CTRL_TEXT);
    GORDER_MOM_STEP(0x111, 0xf, "row_shr:1 row_mask:0xf bank_mask:0xf")
    GORDER_MOM_STEP(0x112, 0xf, "row_shr:2 row_mask:0xf bank_mask:0xf")
    GORDER_MOM_STEP(0x114, 0xf, "row_shr:4 row_mask:0xf bank_mask:0xf")
    GORDER_MOM_STEP(0x118, 0xf, "row_shr:8 row_mask:0xf bank_mask:0xf")
    GORDER_MOM_STEP(0x142, 0xa, "row_bcast:15 row_mask:0xa bank_mask:0xf")
    GORDER_MOM_STEP(0x143, 0xc, "row_bcast:31 row_mask:0xc bank_mask:0xf")
#undef GORDER_MOM_STEP
    const bool all_finite = __all(finite);
    if (lane == 63u) a.mom[(size_t)f * n_tiles + tile_id] = make_float4(all_finite ? s : __builtin_nanf(""), q, mn, mx);
    // the heads this tile owns (at most 64, one a lane, fetched once per workgroup): their normal coordinate out of LDS —
    // k_spec_check would otherwise fetch a cache line per head
    if (my_head.y != 0xffffffffu) a.head_z[(size_t)f * a.n_mol_total + my_head.y] = slot[sh + a.mom_dim + 3u * my_head.x];
}
template <int G, int NPF, bool ACOS_COS, bool PBC, bool LEAF, int AXIS, bool MOM = false>
__global__ __launch_bounds__(kBlock, GORDER_TILED_MIN_WAVES) void k_bonds_tiled(FrameArgs a_in, const float *__restrict__ xyz,
                                                      const float *__restrict__ box9,
                                                      const uint8_t *__restrict__ aflags,
                                                      const uint32_t *__restrict__ arow,
                                                      const Tile *__restrict__ tiles,
                                                      const Item *__restrict__ items,
                                                      const uint32_t *__restrict__ tile_slots,
                                                      uint32_t n_tiles, uint32_t lw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using S = TiledStage<G, NPF, ACOS_COS, PBC, LEAF, AXIS>;
    // the read-only streams come in as __restrict__ kernel arguments so that the compiler can prove
    // that the accumulator / error stores never clobber them (uniform loads become scalar loads)
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles;
    const uint32_t chunk = blockIdx.x / n_tiles;
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const uint32_t sk = tid / S::TPF, si = tid % S::TPF;   // staging role: frame slot, first float4
    const bool active = tid < t.n_items;
    Item it{0, 0, 0, 0, 0};
    if (active) it = items[t.item0 + tid];

    const uint32_t f_begin = chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    const uint32_t f_full = f_begin + ((f_end - f_begin) / G) * G;   // end of the whole stages

    SampleAcc acc;
    int bad = 0;
    uint32_t nan_which = 0, nan_frame = kNoNan;
    v4f pre[NPF];
    uint2 own = make_uint2(0u, 0u), my_head = make_uint2(0u, 0xffffffffu);
    if (MOM) {
        own = a.own[tile_id];
        const uint32_t hq = a.own_head_begin[tile_id] + (tid & 63u);           // (every wave holds the tile's heads, a lane each)
        if (hq < a.own_head_begin[tile_id + 1u]) my_head = a.own_heads[hq];
    }

    if (f_begin < f_full) S::template load<false>(a, t, f_begin, f_end, sk, si, pre);
    for (uint32_t f0 = f_begin; f0 < f_full; f0 += G) {
        const bool finite = S::template store<false, MOM>(a, t, f0, f_end, sk, si, pre, lds, lw);
        __syncthreads();
        if (f0 + G < f_full) S::template load<false>(a, t, f0 + G, f_end, sk, si, pre);   // next stage in flight
        if (MOM) {          // (the wave that staged frame slot k sums it: TPF = 64; before or behind the loads: measured equal)
            static_assert(!MOM || (uint32_t)G * 64u == kBlock, "a wave per frame slot");
            tiled_moments(a, t, own, tile_id, n_tiles, f0 + sk, lds + (size_t)sk * lw, finite, my_head);
        }
        if (active) S::compute(a, t, it, f0, lds, lw, acc, bad, nan_which, nan_frame);
        __syncthreads();
    }
    if (f_full < f_end) {   // last, partial stage of the batch
        S::template load<true>(a, t, f_full, f_end, sk, si, pre);
        const bool finite = S::template store<true, MOM>(a, t, f_full, f_end, sk, si, pre, lds, lw);
        __syncthreads();
        if (MOM && f_full + sk < f_end) tiled_moments(a, t, own, tile_id, n_tiles, f_full + sk, lds + (size_t)sk * lw, finite, my_head);
        if (active) S::compute_tail(a, t, it, f_full, f_end, lds, lw, acc, bad, nan_which, nan_frame);
        __syncthreads();
    }

    if (nan_frame != kNoNan)
        raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_frame, kStageTypes, tile_slots[t.slot0 + it.lslot], 1, it.mol, nan_which);
    if (bad) raise_box_range(a.err, f_begin);

    // ---- epilogue: fold the block's samples per accumulator slot in LDS, then one global atomic
    // per (slot, field).  Integer sums: the result does not depend on the order (order.rs:44-60).
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
#ifdef GORDER_DEBUG_NOEPILOGUE   // timing experiment only
    if (a.n_frames == 0xffffffffu)
#endif
    if (tid < t.n_slots && l_n[tid]) {
        // spread the blocks over n_rep replicas of the accumulator block: same-address atomics of
        // thousands of blocks would otherwise serialise in L2
        unsigned long long *acc = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
        const uint32_t slot = tile_slots[t.slot0 + tid];
        atomicAdd(&acc[slot], l_s[tid]);
        atomicAdd(&acc[2u * a.n_acc + slot], (unsigned long long)l_n[tid]);
        if (l_n[kBlock + tid]) {
            atomicAdd(&acc[a.n_acc + slot], l_s[kBlock + tid]);
            atomicAdd(&acc[3u * a.n_acc + slot], (unsigned long long)l_n[kBlock + tid]);
        }
    }
}

// ---- K1g: same tiles, but every lane gathers its two atoms straight from global memory (through the
// per-CU vector L1) instead of going through an LDS-staged window: no LDS traffic and no barriers in
// the frame loop, waves run fully decoupled.  Each HBM byte is still fetched about once: the lanes
// of a wave touch one contiguous ~1 KiB run of the frame and neighbouring waves share only its ends.
// The loads of stage s+1 are issued before the arithmetic of stage s (2 x G x 6 registers).
template <int G, bool ACOS_COS, bool PBC, bool LEAF>
__global__ __launch_bounds__(kBlock) void k_bonds_gather(FrameArgs a_in, const float *__restrict__ xyz,
                                                       const float *__restrict__ box9,
                                                       const uint8_t *__restrict__ aflags,
                                                       const uint32_t *__restrict__ arow,
                                                       const Tile *__restrict__ tiles,
                                                       const Item *__restrict__ items,
                                                       const uint32_t *__restrict__ tile_slots, uint32_t n_tiles) {
    __shared__ unsigned long long l_s[2 * kBlock];
    __shared__ uint32_t l_n[2 * kBlock];
    using S = TiledStage<G, 1, ACOS_COS, PBC, LEAF>;
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles;
    const uint32_t chunk = blockIdx.x / n_tiles;
    const Tile t = tiles[tile_id];
    const uint32_t tid = threadIdx.x;
    const bool active = tid < t.n_items;
    Item it{0, 0, 0, 0, 0};
    if (active) it = items[t.item0 + tid];
    const uint32_t f_begin = chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    const uint32_t f_full = f_begin + ((f_end - f_begin) / G) * G;
    const size_t fstride = (size_t)a.n_atoms * 3u;
    const float *pi = xyz + ((size_t)t.atom0 + it.li) * 3u;
    const float *pj = xyz + ((size_t)t.atom0 + it.lj) * 3u;

    SampleAcc acc;
    int bad = 0;
    uint32_t nan_which = 0, nan_frame = kNoNan;
    float cur[G][6], nxt[G][6];
    auto fetch = [&](float (&P)[G][6], uint32_t f0) {
#pragma unroll
        for (int k = 0; k < G; k++) {
            const float *q1 = pi + (size_t)(f0 + k) * fstride, *q2 = pj + (size_t)(f0 + k) * fstride;
            P[k][0] = q1[0]; P[k][1] = q1[1]; P[k][2] = q1[2];
            P[k][3] = q2[0]; P[k][4] = q2[1]; P[k][5] = q2[2];
        }
    };
    if (active) {
        if (f_begin < f_full) fetch(cur, f_begin);
        for (uint32_t f0 = f_begin; f0 < f_full; f0 += G) {
            const bool more = f0 + G < f_full;
            if (more) fetch(nxt, f0 + G);
            S::template compute_core<G>(a, t, it, f0, cur, acc, bad, nan_which, nan_frame);
            if (more) {
#pragma unroll
                for (int k = 0; k < G; k++)
#pragma unroll
                    for (int c = 0; c < 6; c++) cur[k][c] = nxt[k][c];
            }
        }
        for (uint32_t f = f_full; f < f_end; f++) {
            const float *q1 = pi + (size_t)f * fstride, *q2 = pj + (size_t)f * fstride;
            const float p1x = q1[0], p1y = q1[1], p1z = q1[2], p2x = q2[0], p2y = q2[1], p2z = q2[2];
            if (bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad) && nan_frame == kNoNan) {
                if (p1x != p1x) { nan_which = 0; nan_frame = f; }
                else if (p2x != p2x) { nan_which = 1; nan_frame = f; }
            }
        }
    }
    if (nan_frame != kNoNan)
        raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_frame, kStageTypes, tile_slots[t.slot0 + it.lslot], 1, it.mol, nan_which);
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
#ifdef GORDER_DEBUG_NOEPILOGUE   // timing experiment only
    if (a.n_frames == 0xffffffffu)
#endif
    if (tid < t.n_slots && l_n[tid]) {
        // spread the blocks over n_rep replicas of the accumulator block: same-address atomics of
        // thousands of blocks would otherwise serialise in L2
        unsigned long long *acc = a.rep + (size_t)(blockIdx.x % a.n_rep) * 4u * a.n_acc;
        const uint32_t slot = tile_slots[t.slot0 + tid];
        atomicAdd(&acc[slot], l_s[tid]);
        atomicAdd(&acc[2u * a.n_acc + slot], (unsigned long long)l_n[tid]);
        if (l_n[kBlock + tid]) {
            atomicAdd(&acc[a.n_acc + slot], l_s[kBlock + tid]);
            atomicAdd(&acc[3u * a.n_acc + slot], (unsigned long long)l_n[kBlock + tid]);
        }
    }
}

// ---- K1b: direct gather (samples whose atoms do not fit one LDS window; also the A/B baseline)
template <bool ACOS_COS>
__global__ __launch_bounds__(256) void k_bonds_direct(FrameArgs a, const DirectItem *__restrict__ items,
                                                       uint32_t n_items, uint32_t blocks_per_chunk) {
    const uint32_t chunk = blockIdx.x / blocks_per_chunk;
    const uint32_t q = (blockIdx.x % blocks_per_chunk) * blockDim.x + threadIdx.x;
    if (q >= n_items) return;
    const DirectItem it = items[q];
    const uint32_t f_begin = chunk * a.frames_per_chunk;
    const uint32_t f_end = min(a.n_frames, f_begin + a.frames_per_chunk);
    SampleAcc acc;
    int bad = 0;
    for (uint32_t f = f_begin; f < f_end; f++) {
        const float *p1 = a.xyz + ((size_t)f * a.n_atoms + it.i) * 3u;
        const float *p2 = a.xyz + ((size_t)f * a.n_atoms + it.j) * 3u;
        const float p1x = p1[0], p1y = p1[1], p1z = p1[2];
        const float p2x = p2[0], p2y = p2[1], p2z = p2[2];
        if (__builtin_expect(bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad), 0)) {
            if (p1x != p1x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, it.slot, 1, it.mol, 0);
            else if (p2x != p2x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, f, kStageTypes, it.slot, 1, it.mol, 1);
        }
    }
    if (bad) raise_box_range(a.err, f_begin);
    if (acc.n_tot) {
        atomicAdd(&a.acc[it.slot], (unsigned long long)acc.s_tot);
        atomicAdd(&a.acc[2u * a.n_acc + it.slot], (unsigned long long)acc.n_tot);
        if (acc.n_up) {
            atomicAdd(&a.acc[a.n_acc + it.slot], (unsigned long long)acc.s_up);
            atomicAdd(&a.acc[3u * a.n_acc + it.slot], (unsigned long long)acc.n_up);
        }
    }
}

}  // namespace
