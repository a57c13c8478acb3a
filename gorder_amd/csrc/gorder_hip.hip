// gorder_hip.hip — MI355X (gfx950) kernels + C ABI of the lipid-order engine.
//
// Path accelerated (reference file:line under /root/reference/src/analysis):
//   analyze_frame                      common.rs:201-235
//   MoleculeTypes::analyze_frame       topology/molecule.rs:54-95
//   BondType::analyze_frame            topology/bond.rs:396-446      -> k_bonds_tiled / k_bonds_direct
//   BondLike::add_order                topology/bond.rs:184-215         (register accumulators)
//   calc_sch / vector_to               mod.rs:78-82, pbc.rs:378-385     (gm_math.h)
//   OrderValue / AnalysisOrder         order.rs:13-66, 178-188          (i64 ticks, u64 counts)
//   check_box                          common.rs:186-198             -> k_check_box
//   SystemLeafletClassification::run   leaflets.rs:171-205           -> k_leaflets_global
//   common_identify_leaflet            leaflets.rs:711-732              (same kernel)
//   IndividualClassification           leaflets.rs:777-801           -> k_leaflets_individual
//   LocalClassification + local centres leaflets.rs:661-675, pbc.rs:273-318 -> k_local_{bin,scan,scatter,flags}
//   should_assign / get_assigned       leaflets.rs:435-441, 1437-1472   (host: assignment-row table)
//   SystemTopology::add / reduce       topology/mod.rs:236-272          (integer sums: order-free)
//
// Built for gfx950 only.  No CPU fallback: without a device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/gorder_hip.h"
#include "gm_math.h"
#include "plan.h"

#pragma clang fp contract(off)

using gorder::DirectItem;
using gorder::Item;
using gorder::kBlock;
using gorder::Plan;
using gorder::Tile;

namespace {

constexpr int kFramesPerStage = 4;   // G: frames staged in LDS per barrier pair (= waves per block)
constexpr uint32_t kErrWords = 4;    // device error record: code, payload, frame, spare

struct FrameArgs {
    const float *xyz;        // [n_frames][n_atoms][3]
    const float *box9;       // [n_frames][9]
    uint32_t n_atoms;
    uint32_t n_frames;       // end of the frame range this launch covers
    uint32_t frame0;         // its begin (only the scatter kernels launch sub-ranges; 0 elsewhere)
    uint32_t frames_per_chunk;
    int pbc;
    float nx, ny, nz, n2, n2sq;   // static normal, its norm and squared norm
    int leaflets;            // 0/1
    const uint8_t *aflags;   // [rows][n_mol_total]
    const uint32_t *arow;    // [n_frames] assignment row of each frame
    uint32_t n_mol_total;
    unsigned long long *acc; // [4][n_acc]: sum_total, sum_upper, cnt_total, cnt_upper
    unsigned long long *rep; // [n_rep][4][n_acc] replicas the tiled kernels add into (folded into acc later)
    uint32_t n_rep;
    uint32_t n_acc;
    uint32_t *err;
};

__device__ __forceinline__ void raise_error(uint32_t *err, uint32_t code, uint32_t payload, uint32_t frame) {
    if (atomicCAS(&err[0], 0u, code) == 0u) {
        err[1] = payload;
        err[2] = frame;
    }
}

// ---- check_box (common.rs:186-198), one thread per frame -----------------------------------
__global__ void k_check_box(const float *__restrict__ box9, uint32_t n_frames, uint32_t *err) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const float *b = box9 + 9 * (size_t)f;
    bool all_nan = true;
    for (int i = 0; i < 9; i++) all_nan = all_nan && (b[i] != b[i]);
    if (all_nan) { raise_error(err, GORDER_ERR_UNDEFINED_BOX, 0, f); return; }
    if (b[1] != 0.0f || b[2] != 0.0f || b[3] != 0.0f || b[5] != 0.0f || b[6] != 0.0f || b[7] != 0.0f) {
        raise_error(err, GORDER_ERR_NOT_ORTHOGONAL_BOX, 0, f);
        return;
    }
    if (b[0] == 0.0f && b[4] == 0.0f && b[8] == 0.0f) { raise_error(err, GORDER_ERR_ZERO_BOX, 0, f); return; }
    if (!(b[0] > 0.0f) || !(b[4] > 0.0f) || !(b[8] > 0.0f)) raise_error(err, GORDER_ERR_BOX_RANGE, 0, f);
}

// total_frames (topology/mod.rs:141-144) lives in the last word of the accumulator block so that a
// multi-GPU all-reduce sums it together with the order sums (topology/mod.rs:243)
__global__ void k_count_frames(unsigned long long *word, uint32_t n_frames) { atomicAdd(word, (unsigned long long)n_frames); }

// acc[i] += sum_r rep[r][i]; rep := 0   (i < 4 * n_acc)
__global__ void k_fold_replicas(unsigned long long *acc, unsigned long long *rep, uint32_t n_rep, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long s = 0;
    for (uint32_t r = 0; r < n_rep; r++) {
        s += rep[(size_t)r * n + i];
        rep[(size_t)r * n + i] = 0;
    }
    acc[i] += s;
}

// ---- ordermap words --------------------------------------------------------------------------
// The scatter kernels add (1 << 42) + tick into one 64-bit word per (plane, slot, tile): the low 42 bits
// hold the signed tick sum, the bits above the sample count.  |tick| <= 1e6, so the sum of c samples stays
// inside 42 signed bits while c < 2^21; the host folds the words into the i64 sum / u64 count maps before
// any tile can have received that many samples (gorder_hip_handle::map_pending).
constexpr unsigned long long kMapOne = 1ull << 42;
constexpr unsigned long long kMapFoldLimit = 1ull << 21;
__device__ __forceinline__ void map_unpack(unsigned long long w, long long &sum, unsigned long long &cnt) {
    sum = (long long)(w << 22) >> 22;                      // sign-extend the low 42 bits
    cnt = (w - (unsigned long long)sum) >> 42;
}
constexpr unsigned long long kMapNoSample = ~0ull;   // staged entry of a lane without a sample in the map
// packed [planes][n] -> sums/cnts [3][n] (total, upper, lower); planes = 2 with leaflets (upper, lower), else 1
__global__ void k_fold_maps(unsigned long long *__restrict__ packed, unsigned long long *__restrict__ sums,
                            unsigned long long *__restrict__ cnts, size_t n, int leaflets) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long w0 = packed[i], w1 = leaflets ? packed[n + i] : 0ull;
        if (!(w0 | w1)) continue;
        long long s0, s1 = 0;
        unsigned long long c0, c1 = 0;
        map_unpack(w0, s0, c0);
        if (leaflets) map_unpack(w1, s1, c1);
        sums[i] += (unsigned long long)(s0 + s1);
        cnts[i] += c0 + c1;
        if (leaflets) {
            if (w0) { sums[n + i] += (unsigned long long)s0; cnts[n + i] += c0; packed[i] = 0; }
            if (w1) { sums[2 * n + i] += (unsigned long long)s1; cnts[2 * n + i] += c1; packed[n + i] = 0; }
        } else {
            packed[i] = 0;
        }
    }
}

// ---- one bond sample (bond.rs:407-443) -----------------------------------------------------
struct SampleAcc {
    long long s_tot = 0, s_up = 0;
    uint32_t n_tot = 0, n_up = 0;
};

// returns true when S came out NaN (undefined position or a non-finite coordinate)
template <bool ACOS_COS>
__device__ __forceinline__ bool bond_sample(const FrameArgs &a, uint32_t f, float p1x, float p1y, float p1z,
                                            float p2x, float p2y, float p2z, uint32_t mol, SampleAcc &acc,
                                            int &bad) {
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
    if (a.leaflets) {
        const uint8_t fl = a.aflags[(size_t)a.arow[f] * a.n_mol_total + mol];
        if (fl == 0) {   // Leaflet::Upper = 0 (lib.rs:416-422)
            acc.s_up += tick;
            acc.n_up += 1;
        }
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
    template <bool TAIL>
    static __device__ __forceinline__ void store(const FrameArgs &a, const Tile &t, uint32_t f0, uint32_t f_end,
                                                 uint32_t sk, uint32_t si, const v4f (&pre)[NPF], float *lds,
                                                 uint32_t lw) {
        const uint32_t f = f0 + sk;
        if (TAIL && f >= f_end) return;
        const size_t base = ((size_t)f * a.n_atoms + t.atom0) * 3u;
        const uint32_t n4 = ((uint32_t)(base & 3u) + 3u * t.n_window + 3u) >> 2;
        const v4f *src = reinterpret_cast<const v4f *>(a.xyz + (base & ~(size_t)3));
        v4f *dst = reinterpret_cast<v4f *>(lds + (size_t)sk * lw);
#pragma unroll
        for (int j = 0; j < NPF; j++) {
            const uint32_t i = si + (uint32_t)j * TPF;
            if (i < n4) dst[i] = pre[j];
        }
        for (uint32_t i = si + (uint32_t)NPF * TPF; i < n4; i += TPF) dst[i] = __builtin_nontemporal_load(src + i);
    }
    // My sample in each of the G frames of a stage; P[k] = {p1x,p1y,p1z,p2x,p2y,p2z} of frame f0 + k.
    // The common path is straight-line code (selects only) so that the G independent dependency chains
    // interleave; the rare cases (atoms more than 1.5 box lengths apart -> literal minimum-image loops;
    // NaN result -> which atom is undefined?) are collected in a bit mask and handled after the stage.
    static __device__ __forceinline__ void compute_core(const FrameArgs &a, const Tile &t, const Item &it,
                                                        uint32_t f0, const float (&P)[G][6], SampleAcc &acc,
                                                        int &bad, uint32_t &nan_atom, uint32_t &nan_frame) {
        int tick[G];
        uint8_t fl[G];
        float bx[G], by[G], bz[G];
        uint32_t rare = 0;
        // uniform per-frame inputs of the whole stage first (scalar loads, issued back to back)
#pragma unroll
        for (int k = 0; k < G; k++) {
            if (PBC) {
                const float *b = a.box9 + 9 * (size_t)(f0 + k);
                bx[k] = b[0]; by[k] = b[4]; bz[k] = b[8];
            }
            if (LEAF) fl[k] = a.aflags[(size_t)a.arow[f0 + k] * a.n_mol_total + it.mol];
        }
#pragma unroll
        for (int k = 0; k < G; k++) {
            float vx = P[k][3] - P[k][0], vy = P[k][4] - P[k][1], vz = P[k][5] - P[k][2];
            bool slow = false;
            if (PBC) {
                vx = gm_min_image_step(vx, bx[k], slow);
                vy = gm_min_image_step(vy, by[k], slow);
                vz = gm_min_image_step(vz, bz[k], slow);
            }
            bool nonfinite = false;
            const float sch = gm_calc_sch<ACOS_COS, AXIS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq, &nonfinite);
            rare |= ((slow || nonfinite || sch != sch) ? 1u : 0u) << k;
            tick[k] = gm_tick(sch);
        }
        if (__builtin_expect(rare != 0, 0)) {
#pragma unroll
            for (int k = 0; k < G; k++) {
                if (!((rare >> k) & 1u)) continue;
                float vx = P[k][3] - P[k][0], vy = P[k][4] - P[k][1], vz = P[k][5] - P[k][2];
                if (PBC) {
                    vx = gm_min_image_loop(vx, bx[k], bad);
                    vy = gm_min_image_loop(vy, by[k], bad);
                    vz = gm_min_image_loop(vz, bz[k], bad);
                }
                const float sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
                tick[k] = gm_tick(sch);
                if (sch != sch) {
                    if (P[k][0] != P[k][0]) { nan_atom = t.atom0 + it.li; nan_frame = f0 + k; }
                    else if (P[k][3] != P[k][3]) { nan_atom = t.atom0 + it.lj; nan_frame = f0 + k; }
                }
            }
        }
        int st = 0, su = 0;
        uint32_t nu = 0;
#pragma unroll
        for (int k = 0; k < G; k++) {
            st += tick[k];
            if (LEAF) {   // Leaflet::Upper = 0 (lib.rs:416-422)
                su += fl[k] == 0 ? tick[k] : 0;
                nu += fl[k] == 0 ? 1u : 0u;
            }
        }
        acc.s_tot += st;
        acc.n_tot += G;
        acc.s_up += su;
        acc.n_up += nu;
    }
    // LDS-staged variant: pick my two atoms out of the staged windows
    static __device__ __forceinline__ void compute(const FrameArgs &a, const Tile &t, const Item &it, uint32_t f0,
                                                   const float *lds, uint32_t lw, SampleAcc &acc, int &bad,
                                                   uint32_t &nan_atom, uint32_t &nan_frame) {
        float P[G][6];
#pragma unroll
        for (int k = 0; k < G; k++) {
            const uint32_t sh = (uint32_t)((((size_t)(f0 + k) * a.n_atoms + t.atom0) * 3u) & 3u);
            const float *w = lds + (size_t)k * lw + sh;
            P[k][0] = w[3u * it.li]; P[k][1] = w[3u * it.li + 1]; P[k][2] = w[3u * it.li + 2];
            P[k][3] = w[3u * it.lj]; P[k][4] = w[3u * it.lj + 1]; P[k][5] = w[3u * it.lj + 2];
        }
        compute_core(a, t, it, f0, P, acc, bad, nan_atom, nan_frame);
    }
    // partial last stage: frames f0 .. f_end-1, one at a time (not performance relevant)
    static __device__ __forceinline__ void compute_tail(const FrameArgs &a, const Tile &t, const Item &it,
                                                        uint32_t f0, uint32_t f_end, const float *lds, uint32_t lw,
                                                        SampleAcc &acc, int &bad, uint32_t &nan_atom,
                                                        uint32_t &nan_frame) {
#pragma unroll 1
        for (uint32_t f = f0; f < f_end; f++) {
            const uint32_t sh = (uint32_t)((((size_t)f * a.n_atoms + t.atom0) * 3u) & 3u);
            const float *w = lds + (size_t)(f - f0) * lw + sh;
            const float p1x = w[3u * it.li], p1y = w[3u * it.li + 1], p1z = w[3u * it.li + 2];
            const float p2x = w[3u * it.lj], p2y = w[3u * it.lj + 1], p2z = w[3u * it.lj + 2];
            if (bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad)) {
                if (p1x != p1x) { nan_atom = t.atom0 + it.li; nan_frame = f; }
                else if (p2x != p2x) { nan_atom = t.atom0 + it.lj; nan_frame = f; }
            }
        }
    }
};

#ifndef GORDER_TILED_MIN_WAVES
#define GORDER_TILED_MIN_WAVES 4   // waves per SIMD the register allocation must allow (8 => <= 64 VGPRs)
#endif
template <int G, int NPF, bool ACOS_COS, bool PBC, bool LEAF, int AXIS>
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
    uint32_t nan_atom = 0xffffffffu, nan_frame = 0;
    v4f pre[NPF];

    if (f_begin < f_full) S::template load<false>(a, t, f_begin, f_end, sk, si, pre);
    for (uint32_t f0 = f_begin; f0 < f_full; f0 += G) {
        S::template store<false>(a, t, f0, f_end, sk, si, pre, lds, lw);
        __syncthreads();
        if (f0 + G < f_full) S::template load<false>(a, t, f0 + G, f_end, sk, si, pre);   // next stage in flight
        if (active) S::compute(a, t, it, f0, lds, lw, acc, bad, nan_atom, nan_frame);
        __syncthreads();
    }
    if (f_full < f_end) {   // last, partial stage of the batch
        S::template load<true>(a, t, f_full, f_end, sk, si, pre);
        S::template store<true>(a, t, f_full, f_end, sk, si, pre, lds, lw);
        __syncthreads();
        if (active) S::compute_tail(a, t, it, f_full, f_end, lds, lw, acc, bad, nan_atom, nan_frame);
        __syncthreads();
    }

    if (nan_atom != 0xffffffffu) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_atom, nan_frame);
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f_begin);

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
    uint32_t nan_atom = 0xffffffffu, nan_frame = 0;
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
            S::compute_core(a, t, it, f0, cur, acc, bad, nan_atom, nan_frame);
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
            if (bond_sample<ACOS_COS>(a, f, p1x, p1y, p1z, p2x, p2y, p2z, it.mol, acc, bad)) {
                if (p1x != p1x) { nan_atom = t.atom0 + it.li; nan_frame = f; }
                else if (p2x != p2x) { nan_atom = t.atom0 + it.lj; nan_frame = f; }
            }
        }
    }
    if (nan_atom != 0xffffffffu) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, nan_atom, nan_frame);
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f_begin);

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
            if (p1x != p1x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, it.i, f);
            else if (p2x != p2x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, it.j, f);
        }
    }
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f_begin);
    if (acc.n_tot) {
        atomicAdd(&a.acc[it.slot], (unsigned long long)acc.s_tot);
        atomicAdd(&a.acc[2u * a.n_acc + it.slot], (unsigned long long)acc.n_tot);
        if (acc.n_up) {
            atomicAdd(&a.acc[a.n_acc + it.slot], (unsigned long long)acc.s_up);
            atomicAdd(&a.acc[3u * a.n_acc + it.slot], (unsigned long long)acc.n_up);
        }
    }
}


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
    unsigned long long *map_rec;     // sample staging (k_map_accumulate): [tile][frame - rec_frame0][1 | 3][kBlock] or null
    uint32_t rec_frame0, rec_frames;
    const float4 *dyn;               // dynamic membrane normals [n_frames][n_mol_total] (nx, ny, nz, cloud size) or null
    int tw;                          // timewise on
    unsigned long long *tw_sums;     // [rows][3][n_acc]
    unsigned long long *tw_cnts;     // [rows][3][n_acc]
    unsigned long long tw_row0;      // row of this batch's frame 0
    // united atoms: sin/cos of the construction angles, evaluated on the host with libm like the reference
    float sin_tet, cos_tet, sin_ch3, cos_ch3, sin_half, cos_half;
    // geometry selection (geometry.rs): per-frame shapes [n_frames][8] = anchor xyz, extents xyz, radius, height
    int geom_kind, geom_invert, geom_orient;
    const float *shapes;
};

// groan_rs Rectangular / Cylinder / Sphere ::inside (oracle: inside_shape), XOR invert (geometry.rs:181-189)
__device__ __forceinline__ bool geom_inside(const ExtraArgs &e, const float *sh, float px, float py, float pz,
                                            const float *box, bool pbc, int &bad) {
    const float p[3] = {px, py, pz};
    bool in = true;
    if (e.geom_kind == GORDER_GEOM_CUBOID) {
        for (int d = 0; d < 3; d++) {
            float x = p[d] - sh[d];
            if (pbc) { x = gm_wrap(x, box[d], bad); in = in && (x <= sh[3 + d]); }
            else in = in && (x >= 0.0f) && (x <= sh[3 + d]);
        }
    } else if (e.geom_kind == GORDER_GEOM_CYLINDER) {
        const int o = e.geom_orient, a = (o + 1) % 3, b = (o + 2) % 3;
        float da = p[a] - sh[a], db = p[b] - sh[b], x = p[o] - sh[o];
        if (pbc) { da = gm_min_image(da, box[a], bad); db = gm_min_image(db, box[b], bad); x = gm_wrap(x, box[o], bad); }
        in = (__builtin_sqrtf(da * da + db * db) < sh[6]) && (pbc ? true : (x >= 0.0f)) && (x <= sh[7]);
    } else {
        float d[3];
        for (int k = 0; k < 3; k++) { d[k] = p[k] - sh[k]; if (pbc) d[k] = gm_min_image(d[k], box[k], bad); }
        in = __builtin_sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) < sh[6];
    }
    return in != (e.geom_invert != 0);
}

// groan_rs GridMap::get_mut_at: nearest tile centre, None outside (oracle: gridmap_index)
__device__ __forceinline__ int grid_index(float x, float lo, float bin, uint32_t n) {
    const float k = __builtin_roundf((x - lo) / bin);
    if (!(k >= 0.0f) || !(k < (float)n)) return -1;
    return (int)k;
}

// BondLike::add_order for the scatter targets (bond.rs:184-215): maps and the per-frame LDS partials
__device__ __forceinline__ void extras_add(const FrameArgs &a, const ExtraArgs &e, uint32_t gslot, uint32_t lslot,
                                           int tick, float px, float py, float pz, int leaflet /* -1 none */,
                                           int *l_tw, uint32_t *l_twn, uint32_t lstride,
                                           unsigned long long *rec = nullptr) {
    if (e.maps) {
        float x, y;
        if (e.plane == 0) { x = px; y = py; }
        else if (e.plane == 1) { x = px; y = pz; }
        else { x = pz; y = py; }
        const int ix = grid_index(x, e.x0, e.binx, e.nx), iy = grid_index(y, e.y0, e.biny, e.ny);
        if (ix >= 0 && iy >= 0) {
            // ONE atomic per sample: count and tick sum share a 64-bit word, and with leaflets only the
            // sample's own leaflet plane is touched (total = upper + lower, bond.rs:199-213); k_fold_maps
            // unpacks.  Scattered 64-bit atomics run at ~24 G/s on gfx950 whatever the scope or table size
            // (tools/microbench/atomic_scatter.hip), so their number is what counts.
            const size_t nt = (size_t)e.nx * e.ny, t = (size_t)ix * e.ny + (size_t)iy;
            if (rec) {   // staged: (plane * tiles + tile) << 32 | tick, added to the map by k_map_accumulate
                *rec = ((unsigned long long)((leaflet > 0 ? nt : 0) + t) << 32) | (unsigned long long)(uint32_t)tick;
            } else {
                const size_t w = leaflet > 0 ? a.n_acc : 0;
                atomicAdd(&e.map_packed[(w + gslot) * nt + t], kMapOne + (unsigned long long)(long long)tick);
            }
        }
    }
    if (e.tw) {
        atomicAdd(&l_tw[lslot], tick);
        atomicAdd(&l_twn[lslot], 1u);
        if (leaflet >= 0) {
            atomicAdd(&l_tw[(1 + leaflet) * lstride + lslot], tick);
            atomicAdd(&l_twn[(1 + leaflet) * lstride + lslot], 1u);
        }
    }
}

// flush the block's per-frame partial sums to the timewise rows (one frame)
__device__ __forceinline__ void extras_flush_tw(const FrameArgs &a, const ExtraArgs &e, const uint32_t *slots,
                                                uint32_t n_slots, uint32_t f, int *l_tw, uint32_t *l_twn,
                                                uint32_t lstride) {
    for (uint32_t ls = threadIdx.x; ls < n_slots; ls += blockDim.x) {
        const size_t row = ((size_t)e.tw_row0 + f) * 3u * a.n_acc;
        for (uint32_t w = 0; w < 3; w++) {
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

template <bool ACOS_COS>
__global__ __launch_bounds__(kBlock) void k_bonds_extras(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                          const float *__restrict__ box9,
                                                          const uint8_t *__restrict__ aflags,
                                                          const uint32_t *__restrict__ arow,
                                                          const Tile *__restrict__ tiles,
                                                          const Item *__restrict__ items,
                                                          const uint32_t *__restrict__ tile_slots, uint32_t n_tiles) {
    __shared__ unsigned long long l_s[2 * kBlock];
    __shared__ uint32_t l_n[2 * kBlock];
    __shared__ int l_tw[3 * kBlock];
    __shared__ uint32_t l_twn[3 * kBlock];
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
    for (uint32_t k = tid; k < 3 * kBlock; k += kBlock) { l_tw[k] = 0; l_twn[k] = 0; }
    __syncthreads();
    SampleAcc acc;
    int bad = 0;
    for (uint32_t f = f_begin; f < f_end; f++) {
        unsigned long long rec = kMapNoSample;
        if (active) {
            const float *q1 = pi + (size_t)f * fstride, *q2 = pj + (size_t)f * fstride;
            const float p1x = q1[0], p1y = q1[1], p1z = q1[2];
            float vx = q2[0] - p1x, vy = q2[1] - p1y, vz = q2[2] - p1z;
            if (a.pbc) {
                const float *b = a.box9 + 9 * (size_t)f;
                vx = gm_min_image(vx, b[0], bad);
                vy = gm_min_image(vy, b[4], bad);
                vz = gm_min_image(vz, b[8], bad);
            }
            if (p1x != p1x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.li, f);
            else if (q2[0] != q2[0]) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.lj, f);
            // bond position = p1 + v / 2 (bond.rs:422); geometry filter (bond.rs:424-426)
            const float mx = p1x + vx / 2.0f, my = p1y + vy / 2.0f, mz = p1z + vz / 2.0f;
            bool in = true;
            if (e.geom_kind) {
                float box[3] = {1.0f, 1.0f, 1.0f};
                if (a.pbc) { const float *b = a.box9 + 9 * (size_t)f; box[0] = b[0]; box[1] = b[4]; box[2] = b[8]; }
                in = geom_inside(e, e.shapes + 8 * (size_t)f, mx, my, mz, box, a.pbc != 0, bad);
            }
            if (in) {
                float sch;
                if (e.dyn) {   // the molecule's own normal of this frame, fetched after the geometry test (bond.rs:429-431)
                    const float4 n = e.dyn[(size_t)f * a.n_mol_total + it.mol];
                    if (n.w < 3.0f) raise_error(a.err, GORDER_ERR_DYNAMIC_NORMAL, (uint32_t)n.w, f);
                    const float n2sq = (n.x * n.x + n.y * n.y) + n.z * n.z;
                    sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, n.x, n.y, n.z, __builtin_sqrtf(n2sq), n2sq);
                } else {
                    sch = gm_calc_sch<ACOS_COS>(vx, vy, vz, a.nx, a.ny, a.nz, a.n2, a.n2sq);
                }
                const int tick = gm_tick(sch);
                int leaflet = -1;
                if (a.leaflets) leaflet = a.aflags[(size_t)a.arow[f] * a.n_mol_total + it.mol] ? 1 : 0;
                acc.s_tot += tick;
                acc.n_tot += 1;
                if (leaflet == 0) { acc.s_up += tick; acc.n_up += 1; }
                extras_add(a, e, gslot, it.lslot, tick, mx, my, mz, leaflet, l_tw, l_twn, kBlock, e.map_rec ? &rec : nullptr);
            }
        }
        if (e.map_rec) e.map_rec[((size_t)tile_id * e.rec_frames + (f - e.rec_frame0)) * kBlock + tid] = rec;
        if (e.tw) {
            __syncthreads();
            extras_flush_tw(a, e, tile_slots + t.slot0, t.n_slots, f, l_tw, l_twn, kBlock);
            __syncthreads();
        }
    }
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f_begin);
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
};
struct PbcLoop {
    V3 box;
    bool pbc;
    bool slow = false;
    int bad = 0;
    __device__ __forceinline__ float len(int k) const { return k == 0 ? box.x : (k == 1 ? box.y : box.z); }
    __device__ __forceinline__ float mi(float d, int k) { return pbc ? gm_min_image_loop(d, len(k), bad) : d; }
    __device__ __forceinline__ float wr(float x, int k) { return pbc ? gm_wrap(x, len(k), bad) : x; }
};
template <typename PB>
__device__ __forceinline__ V3 v3_to(V3 p1, V3 p2, PB &pb) {
    return {pb.mi(p2.x - p1.x, 0), pb.mi(p2.y - p1.y, 1), pb.mi(p2.z - p1.z, 2)};
}
template <typename PB>
__device__ __forceinline__ V3 v3_shift_wrap(V3 t, V3 dir, PB &pb) {
    const V3 u = v3_unit(dir);
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
        const V3 ua = v3_unit(v3_cross(th2, th1));
        const V3 hv1 = v3_rotate(ua, e.sin_tet, e.cos_tet, th1);
        h0 = v3_shift_wrap(target, hv1, pb);
        const V3 n1 = v3_unit(th1);
        h1 = v3_shift_wrap(target, v3_rotate(n1, e.sin_ch3, e.cos_ch3, hv1), pb);
        h2 = v3_shift_wrap(target, v3_rotate(n1, -e.sin_ch3, e.cos_ch3, hv1), pb);
    } else if (kind == GORDER_UA_CH2) {     // uaorder.rs:985-1020
        const V3 th1 = v3_unit(v3_to(target, c.p0, pb)), th2 = v3_unit(v3_to(target, c.p2, pb));
        const V3 pn = v3_cross(th2, th1);
        const V3 ra = v3_unit(V3{th1.x - th2.x, th1.y - th2.y, th1.z - th2.z});
        const V3 rv = v3_cross(pn, ra);
        const V3 ura = v3_unit(ra);
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
        float sn, cs;
        sincosf(ang, &sn, &cs);
        const V3 ua = v3_unit(v3_cross(th1, th2));
        h0 = v3_shift_wrap(target, ang == 0.0f ? th2 : v3_rotate(ua, sn, cs, th2), pb);
    } else {                                // CH1 saturated, uaorder.rs:1087-1104 (h1, h2, h3, target)
        target = c.p3;
        const V3 t1 = v3_unit(v3_to(target, c.p0, pb)), t2 = v3_unit(v3_to(target, c.p1, pb)),
                 t3 = v3_unit(v3_to(target, c.p2, pb));
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
// the literal-loop variant, kept out of line: it runs only for carbons more than 1.5 box lengths away
// from a helper
__device__ __noinline__ UaBonds ua_carbon_slow(uint32_t kind, UaCarbon c, UaConsts e, V3 box, bool pbc) {
    PbcLoop pl{box, pbc};
    return ua_carbon(kind, c, e, pl);
}

template <bool ACOS_COS, bool EXTRAS>
__global__ __launch_bounds__(kBlock) void k_ua_extras(FrameArgs a_in, ExtraArgs e, const float *__restrict__ xyz,
                                                       const float *__restrict__ box9,
                                                       const uint8_t *__restrict__ aflags,
                                                       const uint32_t *__restrict__ arow,
                                                       const Tile *__restrict__ tiles,
                                                       const gorder::UaItem *__restrict__ items,
                                                       const uint32_t *__restrict__ tile_slots, uint32_t n_tiles) {
    constexpr uint32_t LS = 3 * kBlock;   // local slots per block (<= 3 hydrogens per carbon)
    __shared__ unsigned long long l_s[2 * LS];
    __shared__ uint32_t l_n[2 * LS];
    __shared__ int l_tw[EXTRAS ? 3 * LS : 1];
    __shared__ uint32_t l_twn[EXTRAS ? 3 * LS : 1];
    FrameArgs a = a_in;
    a.xyz = xyz; a.box9 = box9; a.aflags = aflags; a.arow = arow;
    const uint32_t tile_id = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
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
    if (EXTRAS)
        for (uint32_t k = tid; k < 3 * LS; k += kBlock) { l_tw[k] = 0; l_twn[k] = 0; }
    for (uint32_t k = tid; k < 2 * LS; k += kBlock) { l_s[k] = 0; l_n[k] = 0; }
    __syncthreads();
    long long s_tot[3] = {0, 0, 0}, s_up[3] = {0, 0, 0};
    uint32_t n_tot[3] = {0, 0, 0}, n_up[3] = {0, 0, 0};
    int bad = 0;
    const bool pbc = a.pbc != 0;
    const UaConsts uc{e.sin_tet, e.cos_tet, e.sin_ch3, e.cos_ch3, e.sin_half, e.cos_half};
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
    for (uint32_t f = f_begin; f < f_end; f++) {
        if (active) {
            const UaCarbon c = fetch(f);
            V3 bx3{1.0f, 1.0f, 1.0f};
            if (pbc) { const float *b = a.box9 + 9 * (size_t)f; bx3 = {b[0], b[4], b[8]}; }
            if (c.p0.x != c.p0.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.l[0], f);
            if (c.p1.x != c.p1.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.l[1], f);
            if (c.p2.x != c.p2.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.l[2], f);
            if (c.p3.x != c.p3.x) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, t.atom0 + it.l[3], f);
            PbcStep ps{bx3, pbc};
            UaBonds ub = ua_carbon(kind, c, uc, ps);
            if (__builtin_expect(ps.slow, 0)) {
                ub = ua_carbon_slow(kind, c, uc, bx3, pbc);
                bad |= ub.bad;
            }
            int leaflet = -1;
            if (a.leaflets) leaflet = a.aflags[(size_t)a.arow[f] * a.n_mol_total + it.mol] ? 1 : 0;
            float nrx = a.nx, nry = a.ny, nrz = a.nz, nr2 = a.n2, nr2sq = a.n2sq;
            if (EXTRAS && e.dyn) {   // fetched for every molecule, before the geometry test (uaorder.rs:412-413)
                const float4 n = e.dyn[(size_t)f * a.n_mol_total + it.mol];
                if (n.w < 3.0f) raise_error(a.err, GORDER_ERR_DYNAMIC_NORMAL, (uint32_t)n.w, f);
                nrx = n.x; nry = n.y; nrz = n.z;
                nr2sq = (n.x * n.x + n.y * n.y) + n.z * n.z;
                nr2 = __builtin_sqrtf(nr2sq);
            }
            unsigned long long recs[3] = {kMapNoSample, kMapNoSample, kMapNoSample};
            auto sample = [&](const int k, const V3 v, const V3 b) {
                if (k >= nh) return;
                const float sch = gm_calc_sch<ACOS_COS>(v.x, v.y, v.z, nrx, nry, nrz, nr2, nr2sq);
                const int tick = gm_tick(sch);
                if (EXTRAS) {
                    const float box[3] = {bx3.x, bx3.y, bx3.z};
                    if (e.geom_kind && !geom_inside(e, e.shapes + 8 * (size_t)f, b.x, b.y, b.z, box, pbc, bad)) return;
                }
                s_tot[k] += tick;
                n_tot[k] += 1;
                if (leaflet == 0) { s_up[k] += tick; n_up[k] += 1; }
                if (EXTRAS)
                    extras_add(a, e, gslot0 + (uint32_t)k, it.lslot0 + (uint32_t)k, tick, b.x, b.y, b.z, leaflet, l_tw, l_twn, LS,
                               e.map_rec ? &recs[k] : nullptr);
            };
            sample(0, ub.v0, ub.b0);
            sample(1, ub.v1, ub.b1);
            sample(2, ub.v2, ub.b2);
            if (EXTRAS && e.map_rec) {   // every lane of the tile writes its three entries: coalesced rows of kBlock words
                unsigned long long *row = e.map_rec + (((size_t)tile_id * e.rec_frames + (f - e.rec_frame0)) * 3u) * kBlock + tid;
                row[0] = recs[0]; row[kBlock] = recs[1]; row[2u * kBlock] = recs[2];
            }
        } else if (EXTRAS && e.map_rec) {
            unsigned long long *row = e.map_rec + (((size_t)tile_id * e.rec_frames + (f - e.rec_frame0)) * 3u) * kBlock + tid;
            row[0] = kMapNoSample; row[kBlock] = kMapNoSample; row[2u * kBlock] = kMapNoSample;
        }
        if (EXTRAS && e.tw) {
            __syncthreads();
            extras_flush_tw(a, e, tile_slots + t.slot0, t.n_slots, f, l_tw, l_twn, LS);
            __syncthreads();
        }
    }
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f_begin);
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

// ---- ordermaps of the united-atom path, second step -----------------------------------------------
// k_ua_extras stages every sample as (plane-tile << 32 | tick) in tile order (coalesced rows); here a block
// owns ONE accumulator slot for a range of frames: it gathers the slot's samples (runs of consecutive lanes,
// gorder::MapRun), adds them into a packed map held in LDS (ds_add_u64) and flushes the tiles it touched
// into the global packed map with one atomic each.  Scattered global atomics run at ~24 G/s on this chip
// whatever one does (tools/microbench/atomic_scatter.hip); this way their number drops from one per sample to
// at most one per (slot, chunk, tile).
__global__ __launch_bounds__(1024) void k_map_accumulate(const unsigned long long *__restrict__ rec,
                                                         const gorder::MapRun *__restrict__ runs,
                                                         const uint32_t *__restrict__ run_begin, uint32_t n_slots,
                                                         uint32_t rec_frames, uint32_t frames_per_chunk, uint32_t k_max,
                                                         uint32_t n_words /* planes * tiles */, uint32_t n_tiles_map,
                                                         unsigned long long *__restrict__ map_packed, uint32_t n_acc) {
    extern __shared__ unsigned long long l_map[];
    const uint32_t slot = blockIdx.x % n_slots, chunk = blockIdx.x / n_slots;
    const uint32_t r0 = run_begin[slot], r1 = run_begin[slot + 1];
    if (r0 == r1) return;                               // no samples of this kind (bond / united atom) in the slot
    const uint32_t f0 = chunk * frames_per_chunk, f1 = min(rec_frames, f0 + frames_per_chunk);
    for (uint32_t w = threadIdx.x; w < n_words; w += blockDim.x) l_map[w] = 0ull;
    __syncthreads();
    for (uint32_t r = r0; r < r1; r++) {
        const gorder::MapRun run = runs[r];
        const uint32_t total = (f1 - f0) * run.n;
        for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
            const uint32_t f = f0 + i / run.n, j = i % run.n;
            const unsigned long long v = rec[(((size_t)run.tile * rec_frames + f) * k_max + run.k) * kBlock + run.tid0 + j];
            if (v != kMapNoSample)
                atomicAdd(&l_map[(uint32_t)(v >> 32)], kMapOne + (unsigned long long)(long long)(int)(uint32_t)v);
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
};

// cos / sin of 2*pi*u by the hardware v_cos_f32 / v_sin_f32 (argument in revolutions, ~1e-6 absolute
// error).  Used only for the Bai-Breen circular-mean ESTIMATE: the estimate merely anchors the
// minimum-image refinement pass that produces the centre, so its last digits do not matter.
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
        // Refinement = plain centre of the atoms' images nearest to the estimate, summed in f32 in atom
        // order like the reference does: a sample 1 ulp from the shape's surface depends on the last bit
        // of this centre (the golden aa_order_sphere_dynamic.yaml has one), so the order of the sum is
        // part of the result.  One thread per frame does it; reference groups are small (a residue, a
        // protein), and the estimate above only selects the images, its own last bits do not matter.
        if (threadIdx.x == 0) {
            float acc[3] = {0.0f, 0.0f, 0.0f};
            for (uint32_t i = 0; i < g.n_group; i++) {
                const float *p = x + 3u * (size_t)g.group[i];
                for (int d = 0; d < 3; d++)
                    acc[d] += g.pbc ? est[d] + gm_min_image(p[d] - est[d], box[d], bad) : p[d];
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
    if (bad) raise_error(g.err, GORDER_ERR_BOX_RANGE, 0, f);
}

// One block per assignment frame: refined Bai-Breen centre of the membrane group
// (leaflets.rs:186-197 -> groan_rs group_get_center) followed by common_identify_leaflet
// (leaflets.rs:711-732) for every molecule.  Per-thread f32 partial sums are combined in f64 (the
// reference sums f32 sequentially; only the sign of head - centre is consumed).
__global__ __launch_bounds__(1024) void k_leaflets_global(LeafletArgs a) {
    __shared__ double scratch[16];
    __shared__ float s_center;
    const uint32_t f = a.aframes[blockIdx.x];
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
            raise_error(a.err, GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER, 0, f);
            c = __builtin_nanf("");
        }
        s_center = c;
    }
    __syncthreads();
    const float cdim = s_center;
    uint8_t *row = a.aflags + (size_t)(a.row0 + blockIdx.x) * a.n_mol_total;
    const bool last = blockIdx.x + 1 == gridDim.x;
    for (uint32_t m = threadIdx.x; m < a.n_mol_total; m += blockDim.x) {
        const float hp = x[3u * (size_t)a.heads[m] + dn];
        float d = hp - cdim;
        if (a.pbc) d = gm_min_image(d, L, bad);
        row[m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
        if (last && a.adist) a.adist[m] = d;
    }
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f);
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
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f);
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
constexpr uint32_t kLocalMaxCells1D = 128;
constexpr uint32_t kLocalSlab = 32;   // assignment frames processed per launch group

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
    const uint32_t *membrane;
    uint32_t n_membrane;
    uint32_t dim;               // normal
    int flip, pbc;
    float radius;
    float radius_thr;           // local_radius_threshold(radius)
    // scratch, per slab frame
    uint32_t *cell_of;          // [n_slab][n_membrane]
    float *trig;                // [n_slab][n_membrane] float4 records in cell order (see k_local_scatter)
    float *rsn;                 // [n_slab][n_membrane] sin of the normal angle, cell order
    uint32_t *cell_count;       // [n_slab][kLocalMaxCells1D^2 + 1] counts -> starts
    uint32_t *cell_fill;        // [n_slab][kLocalMaxCells1D^2]
    uint32_t *err;
};

// In-plane cell grid of one frame.  A dimension with at least 3 radii of box gets cells of radius / k
// (k = kLocalFine, less when the 128-cell cap or the box says so) and a head looks at the 2k+1 cells
// around its own; a smaller dimension is ONE cell (every atom is a candidate exactly once).  Finer cells
// cut the candidates per head from 9 r^2 (k = 1) towards the disk area pi r^2: k = 4 gives 5.1 r^2.
// The grid is this engine's own pruning device — membership itself is the exact distance test.
constexpr uint32_t kLocalFine = 4;
__device__ __forceinline__ void local_axis(float L, float radius, uint32_t &nc, uint32_t &k) {
    nc = 1; k = 0;
    for (uint32_t kk = kLocalFine; kk >= 1u; kk--) {
        // cells are at least 1.0001 radius / kk wide (floor + margin), so +-kk cells reach one radius even
        // when the wrapped coordinates the cells are made from are off by a rounding error
        const float fine = floorf(L / (radius / (float)kk) * 0.9999f);
        if (fine >= (float)(2u * kk + 1u) && fine <= (float)kLocalMaxCells1D) { nc = (uint32_t)fine; k = kk; return; }
    }
}
__device__ __forceinline__ void local_grid(const LocalArgs &a, const float *box, uint32_t &nca, uint32_t &ncb,
                                           int &da, int &db, uint32_t &ka, uint32_t &kb) {
    da = (int)((a.dim + 1u) % 3u);
    db = (int)((a.dim + 2u) % 3u);
    nca = ncb = 1;   // no periodic images to prune with: one cell holds every atom
    ka = kb = 0;
    if (a.pbc) {
        local_axis(box[da], a.radius, nca, ka);
        local_axis(box[db], a.radius, ncb, kb);
    }
}
__device__ __forceinline__ void local_grid(const LocalArgs &a, const float *box, uint32_t &nca, uint32_t &ncb,
                                           int &da, int &db) {
    uint32_t ka, kb;
    local_grid(a, box, nca, ncb, da, db, ka, kb);
}

__device__ __forceinline__ void frame_box(const LocalArgs &a, uint32_t f, float *box) {
    box[0] = box[1] = box[2] = 1.0f;
    if (a.pbc) {
        const float *b = a.box9 + 9 * (size_t)f;
        box[0] = b[0]; box[1] = b[4]; box[2] = b[8];
    }
}

// in-plane cell of membrane atom i in slab frame s (also stored in cell_of)
__device__ __forceinline__ uint32_t local_cell_of(const LocalArgs &a, uint32_t s, uint32_t f, uint32_t i,
                                                  const float *box, uint32_t nca, uint32_t ncb, int da, int db) {
    const float *p = a.xyz + ((size_t)f * a.n_atoms + a.membrane[i]) * 3u;
    int bad = 0;
    uint32_t ca = 0, cb = 0;
    if (a.pbc) {
        const float wa = gm_wrap(p[da], box[da], bad), wb = gm_wrap(p[db], box[db], bad);
        ca = (uint32_t)fminf(fmaxf(floorf(wa / box[da] * (float)nca), 0.0f), (float)(nca - 1u));
        cb = (uint32_t)fminf(fmaxf(floorf(wb / box[db] * (float)ncb), 0.0f), (float)(ncb - 1u));
    }
    const uint32_t c = ca * ncb + cb;
    a.cell_of[(size_t)s * a.n_membrane + i] = c;
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f);
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
    uint32_t nca, ncb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db);
    const uint32_t ncell = nca * ncb;
    uint32_t *count = a.cell_count + (size_t)s * (kLocalMaxCells1D * kLocalMaxCells1D + 1u);
    const bool lds = ncell <= kLocalLdsCells;   // uniform
    if (lds) {
        for (uint32_t k = threadIdx.x; k < ncell; k += blockDim.x) hist[k] = 0;
        __syncthreads();
    }
    if (i < a.n_membrane) {
        const uint32_t c = local_cell_of(a, s, f, i, box, nca, ncb, da, db);
        if (lds) atomicAdd(&hist[c], 1u);
        else atomicAdd(&count[c], 1u);
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
}

// Places every membrane atom in its cell's run and writes a cell-ordered RECORD next to it so that the
// flags kernel streams contiguous data instead of chasing two indices per candidate:
//   rec[q] = (in-plane a, in-plane b, normal coordinate, cos(2 pi wrap(normal)/L)),  rsn[q] = sin(...)
// Only the normal component of the local centre is consumed (leaflets.rs:725), hence one angle.
__global__ __launch_bounds__(256) void k_local_scatter(LocalArgs a) {
    __shared__ uint32_t hist[kLocalLdsCells];   // per-block count, then the block's base offset in each cell
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t s = blockIdx.y;
    const uint32_t f = a.aframes ? a.aframes[s] : a.frame0 + s;
    float box[3];
    frame_box(a, f, box);
    uint32_t nca, ncb;
    int da, db;
    local_grid(a, box, nca, ncb, da, db);
    const uint32_t ncell = nca * ncb;
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
    const float *p = a.xyz + ((size_t)f * a.n_atoms + a.membrane[i]) * 3u;
    int bad = 0;
    float sn = 0.0f, cs = 0.0f;
    if (a.pbc) fast_sincos_rev(gm_wrap(p[dn], box[dn], bad) / box[dn], &sn, &cs);
    const size_t q = (size_t)s * a.n_membrane + start + rank;
    reinterpret_cast<float4 *>(a.trig)[q] = make_float4(p[da], p[db], p[dn], cs);
    a.rsn[q] = sn;
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


// block = 256 threads = 4 waves = 4 heads; grid = (ceil(n_mol / 4), n_slab).  A head's candidates are
// the records of the (2ka+1) x (2kb+1) cells around its own: per row of cells ONE contiguous run of
// records (two when the run wraps around the box), lanes over the run.
__global__ __launch_bounds__(256) void k_local_flags(LocalArgs a) {
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
    const float4 *rec = reinterpret_cast<const float4 *>(a.trig) + (size_t)s * a.n_membrane;
    const float *rsn = a.rsn + (size_t)s * a.n_membrane;
    const float La = box[da], Lb = box[db], Ln = box[dn];
    const float thr = a.radius_thr;
    // rows (ha - ka .. ha + ka) mod nca; in a row the cells (hb - kb .. hb + kb) mod ncb = runs [b0, b1) and [0, b2)
    const uint32_t n_rows = 2u * ka + 1u, n_cols = 2u * kb + 1u;       // <= nca, ncb by local_axis
    const uint32_t a0 = (ha + nca - ka) % nca, b0 = (hb + ncb - kb) % ncb;
    const uint32_t b1 = min(b0 + n_cols, ncb), b2 = b0 + n_cols - b1;
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
    const uint32_t n_runs = 2u * n_rows;
    uint32_t rq0 = 0, rq1 = 0;
    if (lane < n_runs) {
        const uint32_t row = ((a0 + (lane >> 1)) % nca) * ncb;
        rq0 = (lane & 1u) ? cstart[row] : cstart[row + b0];
        rq1 = (lane & 1u) ? cstart[row + b2] : cstart[row + b1];
    }
    uint32_t n_it = 0, it_base = 0, it_end = 0;
    for (uint32_t r = 0; r < n_runs; r++) {
        const uint32_t q0 = __builtin_amdgcn_readlane(rq0, r), q1 = __builtin_amdgcn_readlane(rq1, r);
        const uint32_t n = (q1 - q0 + 63u) >> 6;
        if (lane >= n_it && lane < n_it + n) { it_base = q0 + 64u * (lane - n_it); it_end = q1; }
        n_it += n;
    }
    const bool flat = n_it <= 64u;       // else (> 4096 candidates): the plain run loops

    // pass 1: members (in-plane minimum-image distance < radius; groan_rs Cylinder::inside), their count
    // and the circular sums of the normal coordinate (PBC) or its plain sum (NoPBC).  The membership
    // of the first 64 candidates of each lane is remembered as a bit mask for pass 2.
    float sc = 0.0f, ss = 0.0f, sp = 0.0f;
    uint32_t cnt = 0, nf = 0, it = 0;
    unsigned long long member = 0ull;
    auto take = [&](const float4 r, const float sn, const uint32_t iter) {
        if (inside(r.x, r.y)) {
            cnt += 1;
            if (iter < 64u) member |= 1ull << iter;
            nf |= (r.z - r.z == 0.0f) ? 0u : 1u;
            if (pbc) { sc += r.w; ss += sn; }
            else sp += r.z;
        }
    };
    if (flat) {
        for (uint32_t it0 = 0; it0 < n_it; it0 += 4u) {
            float4 r[4];
            float sn[4];
            bool v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t iter = min(it0 + u, 63u);
                const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)it_base, (int)iter) + lane;
                v[u] = q < (uint32_t)__builtin_amdgcn_readlane((int)it_end, (int)iter);   // lanes >= n_it hold 0: never
                const uint32_t qc = v[u] ? q : 0u;
                r[u] = rec[qc];
                sn[u] = pbc ? rsn[qc] : 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++)
                if (v[u]) take(r[u], sn[u], it0 + u);
        }
    } else {
        for (uint32_t ia = 0; ia < n_rows; ia++) {
            const uint32_t row = ((a0 + ia) % nca) * ncb;
            for (uint32_t part = 0; part < 2u; part++) {
                const uint32_t q0 = part == 0 ? cstart[row + b0] : cstart[row];
                const uint32_t q1 = part == 0 ? cstart[row + b1] : cstart[row + b2];
                for (uint32_t q = q0 + lane; q < q1; q += 64u, it++) take(rec[q], pbc ? rsn[q] : 0.0f, it);
            }
        }
    }
    const double tcnt = wave_total((double)cnt);
    if (tcnt == 0.0 || __any(nf != 0u)) {
        if (lane == 0) raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, a.heads[m], f);
        return;
    }
    float center;
    if (!pbc) {
        center = (float)(wave_total((double)sp) / tcnt);
    } else {
        const double tc = wave_total((double)sc), ts = wave_total((double)ss);
        const float est = (atan2f(-(float)ts, -(float)tc) + 3.1415927f) / (6.2831855f / Ln);
        // pass 2: refine with the mean minimum-image displacement of the members from the estimate
        float ref = 0.0f;
        if (flat) {
            for (uint32_t it0 = 0; it0 < n_it; it0 += 4u) {
                float pn[4];
                bool in[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t iter = min(it0 + u, 63u);
                    in[u] = it0 + u < 64u && ((member >> iter) & 1ull);
                    const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)it_base, (int)iter) + lane;
                    pn[u] = rec[in[u] ? q : 0u].z;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++)
                    if (in[u]) ref += gm_min_image(pn[u] - est, Ln, bad);
            }
        } else {
            it = 0;
            for (uint32_t ia = 0; ia < n_rows; ia++) {
                const uint32_t row = ((a0 + ia) % nca) * ncb;
                for (uint32_t part = 0; part < 2u; part++) {
                    const uint32_t q0 = part == 0 ? cstart[row + b0] : cstart[row];
                    const uint32_t q1 = part == 0 ? cstart[row + b1] : cstart[row + b2];
                    for (uint32_t q = q0 + lane; q < q1; q += 64u, it++) {
                        bool in;
                        float pn;
                        if (it < 64u) {
                            in = (member >> it) & 1ull;
                            pn = in ? rec[q].z : 0.0f;
                        } else {
                            const float4 r = rec[q];
                            in = inside(r.x, r.y);
                            pn = r.z;
                        }
                        if (in) ref += gm_min_image(pn - est, Ln, bad);
                    }
                }
            }
        }
        center = gm_wrap(est + (float)(wave_total((double)ref) / tcnt), Ln, bad);
    }
    if (lane == 0) {
        if (center != center) {
            raise_error(a.err, GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER, a.heads[m], f);
            return;
        }
        float d = hn_pos - center;
        if (pbc) d = gm_min_image(d, Ln, bad);
        a.aflags[(size_t)(a.row0 + s) * a.n_mol_total + m] = (uint8_t)((d >= 0.0f ? 0 : 1) ^ (a.flip ? 1 : 0));
        if ((int)s == a.write_dist_frame && a.adist) a.adist[m] = d;
    }
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f);
}

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
        if (lane == 0) raise_error(a.err, GORDER_ERR_UNDEFINED_POSITION, a.heads[m], f);
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
    const float4 *rec = reinterpret_cast<const float4 *>(a.trig) + (size_t)s * a.n_membrane;
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
                const float4 r = rec[q];
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
    if (bad) raise_error(a.err, GORDER_ERR_BOX_RANGE, 0, f);
}

}  // namespace

// ============================================================================================
// host side
// ============================================================================================

struct gorder_hip_handle {
    Plan plan;
    gorder_tables_t tables{};           // scalar copy (pointers inside are NOT kept)
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // device tables
    Tile *d_tiles = nullptr;
    Item *d_items = nullptr;
    uint32_t *d_tile_slots = nullptr;
    DirectItem *d_direct = nullptr;
    Tile *d_ua_tiles = nullptr;
    gorder::UaItem *d_ua_items = nullptr;
    uint32_t *d_ua_tile_slots = nullptr;
    // ordermaps [3][n_acc][nx*ny] and timewise rows [cap][3][n_acc]
    uint32_t map_nx = 0, map_ny = 0;
    unsigned long long *d_map_sums = nullptr, *d_map_cnts = nullptr;   // [3][n_acc][nx*ny], folded
    unsigned long long *d_map_packed = nullptr;   // [1 or 2][n_acc][nx*ny], what the kernels add into
    unsigned long long *d_map_rec = nullptr;      // united-atom staging for k_map_accumulate (see ExtraArgs::map_rec)
    size_t map_rec_cap = 0;
    gorder::MapRun *d_ua_runs = nullptr, *d_runs = nullptr;
    uint32_t *d_ua_run_begin = nullptr, *d_run_begin = nullptr;
    Item *d_items_by_slot = nullptr;
    bool map_staged = false;       // the packed map of one slot fits LDS: stage + accumulate instead of one atomic per sample
    uint64_t map_pending = 0;      // upper bound of the samples one packed word may hold since the last fold
    uint64_t map_fold_limit = kMapFoldLimit;   // GORDER_HIP_MAP_FOLD_LIMIT lowers it (tests)
    uint32_t map_max_mol = 1;      // most molecules of one type = most samples per tile and frame
    unsigned long long *d_tw_sums = nullptr, *d_tw_cnts = nullptr;
    uint64_t tw_cap = 0;
    ExtraArgs extra{};
    uint32_t *d_geom_group = nullptr;
    float *d_shapes = nullptr;
    // dynamic membrane normals: cloud + per-molecule heads, cell-list scratch (kLocalSlab frames), normals of the batch
    bool dyn = false;
    uint32_t *d_dyn_cloud = nullptr, *d_dyn_heads = nullptr;
    uint32_t *d_dyn_cell_of = nullptr, *d_dyn_count = nullptr;
    float *d_dyn_rec = nullptr, *d_dyn_rsn = nullptr;
    float4 *d_dyn_normals = nullptr;
    size_t dyn_normals_cap = 0;
    std::vector<float> last_normals;   // [n_mol_total][4] of the last submitted frame
    size_t shapes_cap = 0;
    uint32_t *d_err = nullptr;
    unsigned long long *d_acc = nullptr;   // [4][n_acc] + total_frames
    unsigned long long *d_rep = nullptr;   // [n_rep][4][n_acc] (see k_fold_replicas)
    uint32_t n_rep = 32;
    bool rep_dirty = false;
    bool acc_external = false;
    size_t acc_words = 0;
    // leaflets
    uint32_t *d_heads = nullptr, *d_membrane = nullptr, *d_methyl_begin = nullptr, *d_methyl_atoms = nullptr;
    uint8_t *d_aflags = nullptr;
    size_t aflags_rows = 0;
    float *d_adist = nullptr;
    // Local leaflets scratch (sized for kLocalSlab assignment frames)
    uint32_t *d_lcell_of = nullptr, *d_lcell_count = nullptr, *d_lcell_fill = nullptr, *d_lcell_atoms = nullptr;
    float *d_ltrig = nullptr;
    uint32_t *d_arow = nullptr, *d_aframes = nullptr;
    size_t arow_cap = 0, aframes_cap = 0;
    bool have_assignment = false;
    uint64_t assignment_frame = 0;
    // host staging for submit_host
    float *d_stage_xyz = nullptr, *d_stage_box = nullptr;
    size_t stage_xyz_cap = 0, stage_box_cap = 0;
    float n2 = 1.0f, n2sq = 1.0f;
    int axis = -1;   // 0/1/2 when the static normal is exactly that unit axis (kernel specialisation)
    int frames_per_stage = kFramesPerStage;   // G (2, 4 or 8); GORDER_HIP_FRAMES_PER_STAGE overrides
    bool use_gather = false;                   // GORDER_HIP_KERNEL=gather: L1-gather kernel instead of LDS staging
    uint32_t wg_capacity = 256u * 6u;          // co-resident workgroups of the tiled kernel on this device
    uint32_t wg_target = 0;                    // GORDER_HIP_WG_TARGET: force the workgroup count aimed at
    uint32_t lw = 0;
    size_t lds_bytes = 0;
    uint64_t n_frames = 0;
    uint64_t err_index = 0;
    std::string err_msg;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing;
    double timing_ms = 0.0;
    uint64_t timing_launches = 0;
};

namespace {

int fail(gorder_hip_handle *h, int status, const std::string &msg) {
    if (h) h->err_msg = msg;
    return status;
}

#define HIP_TRY(h, expr)                                                                    \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(h, GORDER_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int upload(gorder_hip_handle *h, T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    if (src.empty()) return GORDER_OK;
    HIP_TRY(h, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_TRY(h, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GORDER_OK;
}

template <typename T>
int ensure(gorder_hip_handle *h, T **buf, size_t *cap, size_t need) {
    if (need <= *cap) return GORDER_OK;
    if (*buf) HIP_TRY(h, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    const size_t n = need + need / 2;
    HIP_TRY(h, hipMalloc((void **)buf, n * sizeof(T)));
    *cap = n;
    return GORDER_OK;
}

bool should_assign(uint32_t frequency, uint64_t frame) {   // leaflets.rs:435-441
    return frequency == 0 ? frame == 0 : (frame % frequency) == 0;
}

int check_device_error(gorder_hip_handle *h) {
    uint32_t e[kErrWords];
    HIP_TRY(h, hipMemcpy(e, h->d_err, sizeof(e), hipMemcpyDeviceToHost));
    if (e[0] == 0) return GORDER_OK;
    h->err_index = e[1];
    char buf[160];
    snprintf(buf, sizeof(buf), "device raised %s (payload %u) in batch frame %u", gorder_hip_strerror((int)e[0]),
             e[1], e[2]);
    h->err_msg = buf;
    return (int)e[0];
}

bool env_flag(const char *name) {
    const char *v = getenv(name);
    return v && *v && strcmp(v, "0") != 0;
}

// Launch the order kernels of one batch (internal).
// fold the replicas into the accumulator block (stream-ordered; cheap: 4 * n_acc threads)
int fold_replicas(gorder_hip_handle *h) {
    if (!h->rep_dirty || !h->d_rep) return GORDER_OK;
    const uint32_t n = 4u * h->plan.n_acc;
    hipLaunchKernelGGL(k_fold_replicas, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_acc, h->d_rep, h->n_rep, n);
    HIP_TRY(h, hipGetLastError());
    h->rep_dirty = false;
    return GORDER_OK;
}

// unpack the ordermap words the scatter kernels filled since the last fold (k_fold_maps)
int fold_maps(gorder_hip_handle *h) {
    if (!h->extra.maps || !h->map_pending) return GORDER_OK;
    const size_t n = (size_t)h->plan.n_acc * h->map_nx * h->map_ny;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fold_maps, dim3(blocks), dim3(256), 0, h->stream, h->d_map_packed, h->d_map_sums, h->d_map_cnts,
                       n, h->tables.leaflets.method != GORDER_LEAFLETS_NONE ? 1 : 0);
    HIP_TRY(h, hipGetLastError());
    h->map_pending = 0;
    return GORDER_OK;
}

// normals of every molecule for the frames of this batch -> h->d_dyn_normals [n_frames][n_mol_total]
int run_dynamic_normals(gorder_hip_handle *h, const FrameArgs &a) {
    int st;
    const uint32_t n_mol = h->plan.n_mol_total;
    if ((st = ensure(h, &h->d_dyn_normals, &h->dyn_normals_cap, (size_t)a.n_frames * n_mol)) != GORDER_OK) return st;
    const gorder_dynamic_normal_t &dn = h->tables.dynamic_normal;
    const size_t ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
    LocalArgs lo{};
    lo.xyz = a.xyz; lo.box9 = a.box9; lo.n_atoms = a.n_atoms;
    lo.n_mol_total = n_mol; lo.heads = h->d_dyn_heads; lo.membrane = h->d_dyn_cloud; lo.n_membrane = dn.n_cloud;
    lo.dim = 2; lo.pbc = a.pbc; lo.radius = dn.radius; lo.radius_thr = local_radius_threshold(dn.radius);
    lo.cell_of = h->d_dyn_cell_of; lo.trig = h->d_dyn_rec; lo.rsn = h->d_dyn_rsn;
    lo.cell_count = h->d_dyn_count; lo.cell_fill = h->d_dyn_count + kLocalSlab * (ncell + 1);
    lo.err = h->d_err; lo.aframes = nullptr; lo.write_dist_frame = -1;
    for (uint32_t done = 0; done < a.n_frames; done += kLocalSlab) {
        const uint32_t ns = std::min(a.n_frames - done, kLocalSlab);
        lo.frame0 = done;
        lo.n_slab = ns;
        HIP_TRY(h, hipMemsetAsync(h->d_dyn_count, 0, kLocalSlab * (2 * ncell + 1) * sizeof(uint32_t), h->stream));
        const dim3 ga((dn.n_cloud + 255) / 256, ns);
        hipLaunchKernelGGL(k_local_bin, ga, dim3(256), 0, h->stream, lo);
        hipLaunchKernelGGL(k_local_scan, dim3(ns), dim3(1024), 0, h->stream, lo);
        hipLaunchKernelGGL(k_local_scatter, ga, dim3(256), 0, h->stream, lo);
        hipLaunchKernelGGL(k_dyn_normals, dim3((n_mol + 3) / 4, ns), dim3(256), 0, h->stream, lo, h->d_dyn_normals);
    }
    HIP_TRY(h, hipGetLastError());
    // keep the last frame's normals for gorder_hip_normals (stream-ordered copy into pageable memory)
    h->last_normals.resize(4 * (size_t)n_mol);
    HIP_TRY(h, hipMemcpyAsync(h->last_normals.data(), h->d_dyn_normals + (size_t)(a.n_frames - 1) * n_mol,
                              4 * sizeof(float) * (size_t)n_mol, hipMemcpyDeviceToHost, h->stream));
    return GORDER_OK;
}

int launch_orders(gorder_hip_handle *h, FrameArgs &a) {
    const Plan &p = h->plan;
    const uint32_t n_tiles = (uint32_t)p.tiles.size();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    HIP_TRY(h, hipEventRecord(e0, h->stream));
    const bool extras = h->extra.maps || h->extra.tw || h->extra.geom_kind || h->dyn;
    if (h->dyn) {
        const int st2 = run_dynamic_normals(h, a);
        if (st2 != GORDER_OK) return st2;
    }
    if (h->extra.geom_kind) {
        int st2;
        if ((st2 = ensure(h, &h->d_shapes, &h->shapes_cap, (size_t)a.n_frames * 8)) != GORDER_OK) return st2;
        const gorder_geometry_t &ge = h->tables.geometry;
        GeomArgs ga{};
        ga.xyz = a.xyz; ga.box9 = a.box9; ga.n_atoms = a.n_atoms; ga.pbc = a.pbc;
        ga.kind = ge.kind; ga.reference = ge.reference; ga.orientation = ge.orientation;
        for (int d = 0; d < 3; d++) { ga.point[d] = ge.point[d]; ga.structure_box[d] = ge.structure_box[d]; }
        for (int d = 0; d < 2; d++) { ga.xdim[d] = ge.xdim[d]; ga.ydim[d] = ge.ydim[d]; ga.zdim[d] = ge.zdim[d]; ga.span[d] = ge.span[d]; }
        ga.radius = ge.radius; ga.group = h->d_geom_group; ga.n_group = ge.n_group;
        ga.shapes = h->d_shapes; ga.err = h->d_err;
        hipLaunchKernelGGL(k_geom_shapes, dim3(a.n_frames), dim3(256), 0, h->stream, ga);
        HIP_TRY(h, hipGetLastError());
    }
    if (n_tiles && !extras) {
        // enough workgroups to fill 256 CUs x 8 blocks, frames split into chunks of whole stages
        // Cut the frame range into chunks of whole stages.  All workgroups do the same amount of work,
        // so the grid should be a whole number of co-resident rounds: exactly one round when the tiles
        // fit, otherwise many short rounds so that the last, partial one costs little.
        const uint32_t G = (uint32_t)h->frames_per_stage;
        const uint32_t n_stages = (a.n_frames + G - 1) / G;
        uint32_t target = h->wg_target ? h->wg_target : h->wg_capacity;
        if (!h->wg_target) target = 12u * h->wg_capacity;   // measured: ~8-12 short rounds beat one long round
        uint32_t n_chunks = std::max(1u, target / n_tiles);
        n_chunks = std::min(n_chunks, std::max(1u, n_stages / 4u));   // >= 4 stages per workgroup
        uint32_t fpc = ((n_stages + n_chunks - 1) / n_chunks) * G;
        n_chunks = (a.n_frames + fpc - 1) / fpc;
        a.frames_per_chunk = fpc;
        const uint64_t grid = (uint64_t)n_tiles * n_chunks;
        if (grid > 0x7fffffffull) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "batch too large");
        const bool ac = (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        const dim3 g((uint32_t)grid), b(kBlock);
#define GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, AX_)                                               \
        hipLaunchKernelGGL((k_bonds_tiled<G_, NPF_, AC_, PBC_, LF_, AX_>), g, b, h->lds_bytes, h->stream, a,     \
                           a.xyz, a.box9, a.aflags, a.arow, h->d_tiles, h->d_items, h->d_tile_slots, n_tiles,   \
                           h->lw)
#define GORDER_LAUNCH_TILED_V(G_, NPF_, AC_, PBC_, LF_)                                                    \
        do {                                                                                                \
            if (!(AC_) && h->axis == 2) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 2);                 \
            else if (!(AC_) && h->axis == 1) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 1);            \
            else if (!(AC_) && h->axis == 0) GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, 0);            \
            else GORDER_LAUNCH_TILED_A(G_, NPF_, AC_, PBC_, LF_, -1);                                       \
        } while (0)
#define GORDER_LAUNCH_TILED(G_, NPF_)                                                                       \
        do {                                                                                                \
            const int v_ = (ac ? 4 : 0) | (a.pbc ? 2 : 0) | (a.leaflets ? 1 : 0);                           \
            switch (v_) {                                                                                   \
                case 0: GORDER_LAUNCH_TILED_V(G_, NPF_, false, false, false); break;                        \
                case 1: GORDER_LAUNCH_TILED_V(G_, NPF_, false, false, true); break;                         \
                case 2: GORDER_LAUNCH_TILED_V(G_, NPF_, false, true, false); break;                         \
                case 3: GORDER_LAUNCH_TILED_V(G_, NPF_, false, true, true); break;                          \
                case 4: GORDER_LAUNCH_TILED_V(G_, NPF_, true, false, false); break;                         \
                case 5: GORDER_LAUNCH_TILED_V(G_, NPF_, true, false, true); break;                          \
                case 6: GORDER_LAUNCH_TILED_V(G_, NPF_, true, true, false); break;                          \
                default: GORDER_LAUNCH_TILED_V(G_, NPF_, true, true, true); break;                          \
            }                                                                                               \
        } while (0)
#define GORDER_LAUNCH_GATHER_V(G_, AC_, PBC_, LF_)                                                         \
        hipLaunchKernelGGL((k_bonds_gather<G_, AC_, PBC_, LF_>), g, b, 0, h->stream, a, a.xyz, a.box9, a.aflags, \
                           a.arow, h->d_tiles, h->d_items, h->d_tile_slots, n_tiles)
#define GORDER_LAUNCH_GATHER(G_)                                                                            \
        do {                                                                                                \
            const int v_ = (ac ? 4 : 0) | (a.pbc ? 2 : 0) | (a.leaflets ? 1 : 0);                           \
            switch (v_) {                                                                                   \
                case 0: GORDER_LAUNCH_GATHER_V(G_, false, false, false); break;                             \
                case 1: GORDER_LAUNCH_GATHER_V(G_, false, false, true); break;                              \
                case 2: GORDER_LAUNCH_GATHER_V(G_, false, true, false); break;                              \
                case 3: GORDER_LAUNCH_GATHER_V(G_, false, true, true); break;                               \
                case 4: GORDER_LAUNCH_GATHER_V(G_, true, false, false); break;                              \
                case 5: GORDER_LAUNCH_GATHER_V(G_, true, false, true); break;                               \
                case 6: GORDER_LAUNCH_GATHER_V(G_, true, true, false); break;                               \
                default: GORDER_LAUNCH_GATHER_V(G_, true, true, true); break;                               \
            }                                                                                               \
        } while (0)
        if (h->use_gather) {
            GORDER_LAUNCH_GATHER(4);
        } else {
            switch (h->frames_per_stage) {
                case 8: GORDER_LAUNCH_TILED(8, 10); break;
                default: GORDER_LAUNCH_TILED(4, 5); break;
            }
        }
#undef GORDER_LAUNCH_GATHER
#undef GORDER_LAUNCH_GATHER_V
#undef GORDER_LAUNCH_TILED_A
#undef GORDER_LAUNCH_TILED_V
#undef GORDER_LAUNCH_TILED
        HIP_TRY(h, hipGetLastError());
    }
    if (extras || !p.ua_tiles.empty()) {
        // scatter-bound modes: plain per-sample kernels (see "Extras" above)
        ExtraArgs e = h->extra;
        e.tw_sums = h->d_tw_sums; e.tw_cnts = h->d_tw_cnts; e.tw_row0 = h->n_frames;
        e.shapes = h->d_shapes;
        e.dyn = h->dyn ? h->d_dyn_normals : nullptr;
        const bool ac = (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        // with ordermaps the frames go in sub-ranges short enough for the packed map words (k_fold_maps)
        uint32_t sub = e.maps ? (uint32_t)std::max<uint64_t>(1, (h->map_fold_limit - 1) / h->map_max_mol) : a.n_frames;
        const bool staged = e.maps && h->map_staged;
        const size_t rec_per_frame = std::max(p.ua_tiles.size() * 3u, extras ? p.tiles.size() : (size_t)0) * kBlock;   // staged words per frame
        if (staged) {   // at most 1 GiB of staging per sub-range
            sub = std::min<uint32_t>(sub, (uint32_t)std::max<size_t>(1, ((size_t)1 << 27) / rec_per_frame));
            sub = std::min(sub, a.n_frames);
            const int st2 = ensure(h, &h->d_map_rec, &h->map_rec_cap, rec_per_frame * sub);
            if (st2 != GORDER_OK) return st2;
        }
        for (uint32_t lo = 0; lo < a.n_frames; lo += sub) {
            const uint32_t hi = std::min(a.n_frames, lo + sub), nf = hi - lo;
            if (e.maps) {
                const uint64_t cost = (uint64_t)nf * h->map_max_mol;
                if (h->map_pending + cost >= h->map_fold_limit) {
                    const int st = fold_maps(h);
                    if (st != GORDER_OK) return st;
                }
                h->map_pending += cost;
            }
            for (int pass = 0; pass < 2; pass++) {
                const uint32_t nt = pass == 0 ? (extras ? n_tiles : 0u) : (uint32_t)p.ua_tiles.size();
                if (!nt) continue;
                uint32_t n_chunks = std::max(1u, (h->wg_target ? h->wg_target : 8u * h->wg_capacity) / nt);
                n_chunks = std::min(n_chunks, nf);
                const uint32_t fpc = (nf + n_chunks - 1) / n_chunks;
                n_chunks = (nf + fpc - 1) / fpc;
                FrameArgs b = a;
                b.frame0 = lo;
                b.n_frames = hi;
                b.frames_per_chunk = fpc;
                e.map_rec = staged ? h->d_map_rec : nullptr;
                e.rec_frame0 = lo;
                e.rec_frames = nf;
                const dim3 g(nt * n_chunks), blk(kBlock);
                if (pass == 0) {
                    const Item *items = staged ? h->d_items_by_slot : h->d_items;
                    if (ac) hipLaunchKernelGGL(k_bonds_extras<true>, g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags,
                                               b.arow, h->d_tiles, items, h->d_tile_slots, nt);
                    else hipLaunchKernelGGL(k_bonds_extras<false>, g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags,
                                            b.arow, h->d_tiles, items, h->d_tile_slots, nt);
                } else {
#define GORDER_LAUNCH_UA(AC, EX)                                                                                  \
    hipLaunchKernelGGL((k_ua_extras<AC, EX>), g, blk, 0, h->stream, b, e, b.xyz, b.box9, b.aflags, b.arow,          \
                       h->d_ua_tiles, h->d_ua_items, h->d_ua_tile_slots, nt)
                    if (extras) { if (ac) GORDER_LAUNCH_UA(true, true); else GORDER_LAUNCH_UA(false, true); }
                    else { if (ac) GORDER_LAUNCH_UA(true, false); else GORDER_LAUNCH_UA(false, false); }
#undef GORDER_LAUNCH_UA
                }
                {
                    if (staged) {   // second step: slot-major accumulation of the staged samples in LDS
                        const uint32_t planes = h->tables.leaflets.method != GORDER_LEAFLETS_NONE ? 2u : 1u;
                        const uint32_t ntm = h->map_nx * h->map_ny, n_words = planes * ntm;
                        // enough blocks for ~2 per CU; a block flushes <= n_words atomics, so keep its chunk long
                        uint32_t mchunks = std::max(1u, 512u / std::max(1u, p.n_acc));
                        mchunks = std::min(mchunks, std::max(1u, nf / 16u));
                        const uint32_t mfpc = (nf + mchunks - 1) / mchunks;
                        mchunks = (nf + mfpc - 1) / mfpc;
                        hipLaunchKernelGGL(k_map_accumulate, dim3(p.n_acc * mchunks), dim3(1024),
                                           n_words * sizeof(unsigned long long), h->stream, h->d_map_rec,
                                           pass == 0 ? h->d_runs : h->d_ua_runs, pass == 0 ? h->d_run_begin : h->d_ua_run_begin,
                                           p.n_acc, nf, mfpc, pass == 0 ? 1u : 3u, n_words, ntm, h->d_map_packed, p.n_acc);
                    }
                }
                HIP_TRY(h, hipGetLastError());
            }
        }
    }
    if (!p.direct.empty()) {
        const uint32_t n_items = (uint32_t)p.direct.size();
        const uint32_t bpc = (n_items + kBlock - 1) / kBlock;
        const uint32_t target = h->wg_target ? h->wg_target : 256u * 8u;
        uint32_t n_chunks = std::max(1u, (target + bpc - 1) / bpc);
        n_chunks = std::min(n_chunks, a.n_frames);
        const uint32_t fpc = (a.n_frames + n_chunks - 1) / n_chunks;
        n_chunks = (a.n_frames + fpc - 1) / fpc;
        FrameArgs b = a;
        b.frames_per_chunk = fpc;
        if (h->tables.flags & GORDER_FLAG_TRIG_ACOS_COS)
            hipLaunchKernelGGL(k_bonds_direct<true>, dim3(bpc * n_chunks), dim3(kBlock), 0, h->stream, b, h->d_direct,
                               n_items, bpc);
        else
            hipLaunchKernelGGL(k_bonds_direct<false>, dim3(bpc * n_chunks), dim3(kBlock), 0, h->stream, b, h->d_direct,
                               n_items, bpc);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL(k_count_frames, dim3(1), dim3(1), 0, h->stream, a.acc + 4 * (size_t)a.n_acc, a.n_frames);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(e1, h->stream));
    h->timing.emplace_back(e0, e1);
    h->timing_launches += 1;
    return GORDER_OK;
}

}  // namespace

extern "C" {

const char *gorder_hip_strerror(int status) {
    switch (status) {
        case GORDER_OK: return "ok";
        case GORDER_ERR_UNDEFINED_BOX: return "system has undefined simulation box";
        case GORDER_ERR_NOT_ORTHOGONAL_BOX: return "the simulation box is not orthogonal";
        case GORDER_ERR_ZERO_BOX: return "all dimensions of the simulation box are zero";
        case GORDER_ERR_UNDEFINED_POSITION: return "atom has an undefined position";
        case GORDER_ERR_INVALID_GLOBAL_MEMBRANE_CENTER: return "could not calculate global membrane center";
        case GORDER_ERR_INVALID_LOCAL_MEMBRANE_CENTER: return "could not calculate local membrane center";
        case GORDER_ERR_DYNAMIC_NORMAL: return "not enough points for dynamic local membrane normal calculation (need 3)";
        case GORDER_ERR_INVALID_ARGUMENT: return "invalid argument";
        case GORDER_ERR_DEVICE: return "HIP runtime error";
        case GORDER_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
        case GORDER_ERR_BOX_RANGE: return "box edge <= 0 or coordinate too far outside the box";
        case GORDER_ERR_LEAFLETS_NOT_PRIMED: return "leaflet assignment missing for the first frame";
        case GORDER_ERR_OVERFLOW: return "order accumulator overflowed";
        default: return "unknown status";
    }
}

int gorder_hip_plan_tables(const gorder_tables_t *tables, gorder_hip_plan_t *out, int *selfcheck) {
    if (!tables || !out) return GORDER_ERR_INVALID_ARGUMENT;
    Plan p;
    const int st = gorder::build_plan(*tables, env_flag("GORDER_HIP_FORCE_DIRECT"), p);
    if (st != GORDER_OK) return st;
    out->n_tiles = (uint32_t)p.tiles.size();
    out->block_threads = kBlock;
    out->max_window_atoms = p.max_window;
    out->n_direct_items = (uint32_t)p.direct.size();
    out->frames_per_stage = kFramesPerStage;
    const uint32_t lw = ((3u * p.max_window + 3u + 3u) / 4u) * 4u;
    size_t lds = (size_t)kFramesPerStage * lw * sizeof(float);
    if (lds < (size_t)kBlock * 24) lds = (size_t)kBlock * 24;
    out->lds_bytes = (uint32_t)lds;
    if (selfcheck) *selfcheck = gorder::selfcheck_plan(*tables, p);
    return GORDER_OK;
}

int gorder_hip_create(const gorder_tables_t *t, gorder_hip_handle **out) {
    if (!t || !out) return GORDER_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return GORDER_ERR_NO_DEVICE;
    if (t->device < 0 || t->device >= n_dev) return GORDER_ERR_INVALID_ARGUMENT;

    gorder_hip_handle *h = new (std::nothrow) gorder_hip_handle();
    if (!h) return GORDER_ERR_DEVICE;
    *out = h;   // returned even on failure so that the caller can read the message; destroy() is safe
    h->tables = *t;
    h->tables.molecule_types = nullptr;
    h->tables.leaflets.membrane = nullptr;
    h->device = t->device;
    HIP_TRY(h, hipSetDevice(h->device));
    int st = gorder::build_plan(*t, env_flag("GORDER_HIP_FORCE_DIRECT"), h->plan);
    if (st != GORDER_OK) return fail(h, st, "invalid bond tables");
    const Plan &p = h->plan;
    HIP_TRY(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    if ((st = upload(h, &h->d_tiles, p.tiles)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_items, p.items)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_tile_slots, p.tile_slots)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_direct, p.direct)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_tiles, p.ua_tiles)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_items, p.ua_items)) != GORDER_OK) return st;
    if ((st = upload(h, &h->d_ua_tile_slots, p.ua_tile_slots)) != GORDER_OK) return st;
    {
        ExtraArgs &e = h->extra;
        // construction angles of the united-atom hydrogens (uaorder.rs:34-41); sin/cos with the host
        // libm, as the reference's nalgebra Rotation3::from_axis_angle does
        e.sin_tet = sinf(1.910633f); e.cos_tet = cosf(1.910633f);
        e.sin_ch3 = sinf(2.0943952f); e.cos_ch3 = cosf(2.0943952f);
        e.sin_half = sinf(0.9553165f); e.cos_half = cosf(0.9553165f);
        const gorder_ordermap_t &om = t->ordermap;
        if (om.enabled) {
            if (!(om.bin[0] > 0.0f) || !(om.bin[1] > 0.0f) || om.plane > 2)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "ordermap bin/plane");
            // groan_rs GridMap::new: n = round(span / bin) + 1 tiles per axis
            const float fx = roundf((om.span_x[1] - om.span_x[0]) / om.bin[0]);
            const float fy = roundf((om.span_y[1] - om.span_y[0]) / om.bin[1]);
            if (!(fx >= 0.0f) || !(fy >= 0.0f) || fx > 65535.0f || fy > 65535.0f)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "ordermap span");
            h->map_nx = (uint32_t)fx + 1u;
            h->map_ny = (uint32_t)fy + 1u;
            const size_t nmap = 3 * (size_t)p.n_acc * h->map_nx * h->map_ny;
            HIP_TRY(h, hipMalloc((void **)&h->d_map_sums, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMalloc((void **)&h->d_map_cnts, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_sums, 0, nmap * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_cnts, 0, nmap * sizeof(unsigned long long)));
            const size_t npk = (t->leaflets.method != GORDER_LEAFLETS_NONE ? 2 : 1) * (nmap / 3);
            HIP_TRY(h, hipMalloc((void **)&h->d_map_packed, npk * sizeof(unsigned long long)));
            HIP_TRY(h, hipMemset(h->d_map_packed, 0, npk * sizeof(unsigned long long)));
            for (uint32_t m = 0; m < t->n_molecule_types; m++)
                h->map_max_mol = std::max(h->map_max_mol, t->molecule_types[m].n_molecules);
            // united atoms: stage + accumulate in LDS when one slot's packed map (x2 with leaflets) fits
            const size_t lds_bytes = npk / p.n_acc * sizeof(unsigned long long);
            if (lds_bytes <= 150u * 1024u && !env_flag("GORDER_HIP_MAP_DIRECT")) {
                if ((st = upload(h, &h->d_ua_runs, p.ua_runs)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_ua_run_begin, p.ua_run_begin)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_runs, p.runs)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_run_begin, p.run_begin)) != GORDER_OK) return st;
                if ((st = upload(h, &h->d_items_by_slot, p.items_by_slot)) != GORDER_OK) return st;
                HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_map_accumulate),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                h->map_staged = true;
            }
            e.maps = 1; e.plane = om.plane; e.x0 = om.span_x[0]; e.y0 = om.span_y[0];
            e.binx = om.bin[0]; e.biny = om.bin[1]; e.nx = h->map_nx; e.ny = h->map_ny;
            e.map_packed = h->d_map_packed;
        }
        e.tw = t->timewise ? 1 : 0;
        const gorder_geometry_t &ge = t->geometry;
        if (ge.kind != GORDER_GEOM_NONE) {
            if (ge.kind > GORDER_GEOM_SPHERE || ge.reference > GORDER_GEOMREF_GROUP || ge.orientation > 2)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry kind/reference/orientation");
            if (ge.reference == GORDER_GEOMREF_BOX_CENTER && !t->handle_pbc)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "box-centre reference needs handle_pbc (pbc.rs:243-245)");
            if (ge.reference == GORDER_GEOMREF_POINT && t->handle_pbc &&
                !(ge.structure_box[0] > 0.0f && ge.structure_box[1] > 0.0f && ge.structure_box[2] > 0.0f))
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry.structure_box");
            if (ge.reference == GORDER_GEOMREF_GROUP) {
                if (!ge.group || ge.n_group == 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry.group");
                std::vector<uint32_t> grp(ge.group, ge.group + ge.n_group);
                for (uint32_t a : grp)
                    if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "geometry group index out of range");
                if ((st = upload(h, &h->d_geom_group, grp)) != GORDER_OK) return st;
            }
            e.geom_kind = (int)ge.kind; e.geom_invert = ge.invert ? 1 : 0; e.geom_orient = (int)ge.orientation;
            h->tables.geometry.group = nullptr;
        }
        if ((e.maps || e.tw || e.geom_kind) && !p.direct.empty())
            return fail(h, GORDER_ERR_INVALID_ARGUMENT,
                        "ordermaps / timewise / geometry need every bond to fit an atom window");
    }
    HIP_TRY(h, hipMalloc((void **)&h->d_err, kErrWords * sizeof(uint32_t)));
    HIP_TRY(h, hipMemset(h->d_err, 0, kErrWords * sizeof(uint32_t)));
    h->acc_words = 4 * (size_t)p.n_acc + 1;
    HIP_TRY(h, hipMalloc((void **)&h->d_acc, h->acc_words * sizeof(unsigned long long)));
    HIP_TRY(h, hipMemset(h->d_acc, 0, h->acc_words * sizeof(unsigned long long)));
    if (const char *e = getenv("GORDER_HIP_MAP_FOLD_LIMIT")) {
        const long v = atol(e);
        if (v >= 2 && (unsigned long long)v <= kMapFoldLimit) h->map_fold_limit = (uint64_t)v;
    }
    if (const char *e = getenv("GORDER_HIP_REPLICAS")) {
        const int r = atoi(e);
        if (r >= 1 && r <= 1024) h->n_rep = (uint32_t)r;
    }
    if (p.n_acc) {
        const size_t rep_bytes = (size_t)h->n_rep * 4u * p.n_acc * sizeof(unsigned long long);
        HIP_TRY(h, hipMalloc((void **)&h->d_rep, rep_bytes));
        HIP_TRY(h, hipMemset(h->d_rep, 0, rep_bytes));
    }
    {   // |normal| with the f32 sequence of nalgebra's norm (oracle: norm3)
        const float *n = t->normal;
        h->n2sq = (n[0] * n[0] + n[1] * n[1]) + n[2] * n[2];
        h->n2 = sqrtf(h->n2sq);
        for (int d = 0; d < 3; d++)
            if (n[d] == 1.0f && n[(d + 1) % 3] == 0.0f && n[(d + 2) % 3] == 0.0f) h->axis = d;
    }
    if (const char *e = getenv("GORDER_HIP_FRAMES_PER_STAGE")) {
        const int g = atoi(e);
        if (g == 4 || g == 8) h->frames_per_stage = g;
    }
    if (const char *e = getenv("GORDER_HIP_KERNEL")) h->use_gather = strcmp(e, "gather") == 0;
    if (const char *e = getenv("GORDER_HIP_WG_TARGET")) {
        const int w = atoi(e);
        if (w > 0) h->wg_target = (uint32_t)w;
    }
    h->lw = ((3u * p.max_window + 3u + 3u) / 4u) * 4u;
    h->lds_bytes = (size_t)h->frames_per_stage * h->lw * sizeof(float);
    if (h->lds_bytes < (size_t)kBlock * 24) h->lds_bytes = (size_t)kBlock * 24;
    {   // how many workgroups of the tiled kernel are co-resident: the frame range of a batch is cut so
        // that the grid is a whole number of such rounds (no half-empty last round)
        int n_cu = 256, per_cu = 6;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
        const bool ac = (t->flags & GORDER_FLAG_TRIG_ACOS_COS) != 0;
        hipError_t e;
        switch (h->frames_per_stage) {
            case 8: e = ac ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<8, 10, true, true, false, -1>, kBlock, h->lds_bytes)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<8, 10, false, true, false, 2>, kBlock, h->lds_bytes); break;
            default: e = ac ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<4, 5, true, true, false, -1>, kBlock, h->lds_bytes)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bonds_tiled<4, 5, false, true, false, 2>, kBlock, h->lds_bytes); break;
        }
        if (e != hipSuccess || per_cu < 1) per_cu = 4;
        h->wg_capacity = (uint32_t)n_cu * (uint32_t)per_cu;
    }

    const gorder_dynamic_normal_t &dn = t->dynamic_normal;
    if (dn.enabled) {
        if (!(dn.radius > 0.0f)) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need a positive radius");
        if (!dn.cloud || dn.n_cloud == 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need the NormalHeads group");
        if (!p.direct.empty()) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals: a bond spans more than the LDS window");
        std::vector<uint32_t> cloud(dn.cloud, dn.cloud + dn.n_cloud), nheads;
        for (uint32_t a : cloud)
            if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "NormalHeads index out of range");
        for (uint32_t m = 0; m < t->n_molecule_types; m++) {
            const gorder_moltype_t &mt = t->molecule_types[m];
            if (!mt.normal_heads) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "dynamic normals need normal_heads[] per molecule type");
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                if (mt.normal_heads[k] >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "normal head index out of range");
                nheads.push_back(mt.normal_heads[k]);
            }
        }
        if ((st = upload(h, &h->d_dyn_cloud, cloud)) != GORDER_OK) return st;
        if ((st = upload(h, &h->d_dyn_heads, nheads)) != GORDER_OK) return st;
        const size_t nm = dn.n_cloud, ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_cell_of, kLocalSlab * nm * sizeof(uint32_t)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_rsn, kLocalSlab * nm * sizeof(float)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_rec, kLocalSlab * nm * 4 * sizeof(float)));
        HIP_TRY(h, hipMalloc((void **)&h->d_dyn_count, kLocalSlab * (2 * ncell + 1) * sizeof(uint32_t)));
        h->dyn = true;
    }

    const gorder_leaflets_t &lf = t->leaflets;
    if (lf.method != GORDER_LEAFLETS_NONE) {
        if (lf.normal_dim > 2) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "leaflets.normal_dim");
        std::vector<uint32_t> heads, mb(1, 0), ma;
        for (uint32_t m = 0; m < t->n_molecule_types; m++) {
            const gorder_moltype_t &mt = t->molecule_types[m];
            if (lf.method != GORDER_LEAFLETS_MANUAL && !mt.heads)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "leaflets need heads[] per molecule type");
            if (lf.method == GORDER_LEAFLETS_INDIVIDUAL && (!mt.methyls || mt.n_methyls == 0))
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "individual leaflets need methyls[]");
            for (uint32_t k = 0; k < mt.n_molecules; k++) {
                const uint32_t hd = mt.heads ? mt.heads[k] : 0;
                if (hd >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "head index out of range");
                heads.push_back(hd);
                if (lf.method == GORDER_LEAFLETS_INDIVIDUAL)
                    for (uint32_t q = 0; q < mt.n_methyls; q++) {
                        const uint32_t a = mt.methyls[(size_t)k * mt.n_methyls + q];
                        if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "methyl index out of range");
                        ma.push_back(a);
                    }
                mb.push_back((uint32_t)ma.size());
            }
        }
        if ((st = upload(h, &h->d_heads, heads)) != GORDER_OK) return st;
        if ((st = upload(h, &h->d_methyl_begin, mb)) != GORDER_OK) return st;
        if ((st = upload(h, &h->d_methyl_atoms, ma)) != GORDER_OK) return st;
        if (lf.method == GORDER_LEAFLETS_LOCAL && !(lf.radius > 0.0f))
            return fail(h, GORDER_ERR_INVALID_ARGUMENT, "local leaflets need a positive radius");
        if (lf.method == GORDER_LEAFLETS_GLOBAL || lf.method == GORDER_LEAFLETS_LOCAL) {
            if (!lf.membrane || lf.n_membrane == 0)
                return fail(h, GORDER_ERR_INVALID_ARGUMENT, "global/local leaflets need the membrane group");
            std::vector<uint32_t> mem(lf.membrane, lf.membrane + lf.n_membrane);
            for (uint32_t a : mem)
                if (a >= t->n_atoms) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "membrane index out of range");
            if ((st = upload(h, &h->d_membrane, mem)) != GORDER_OK) return st;
        }
        if (lf.method == GORDER_LEAFLETS_LOCAL) {
            const size_t nm = lf.n_membrane, ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_of, kLocalSlab * nm * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_atoms, kLocalSlab * nm * sizeof(float)));   // sin column
            HIP_TRY(h, hipMalloc((void **)&h->d_ltrig, kLocalSlab * nm * 4 * sizeof(float)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_count, kLocalSlab * (ncell + 1) * sizeof(uint32_t)));
            HIP_TRY(h, hipMalloc((void **)&h->d_lcell_fill, kLocalSlab * ncell * sizeof(uint32_t)));
        }
        HIP_TRY(h, hipMalloc((void **)&h->d_adist, sizeof(float) * (p.n_mol_total ? p.n_mol_total : 1)));
    }
    // the memsets above ran on the null stream, which the handle's non-blocking stream does not wait for
    HIP_TRY(h, hipDeviceSynchronize());
    return GORDER_OK;
}

void gorder_hip_destroy(gorder_hip_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &ev : h->timing) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    (void)hipFree(h->d_tiles); (void)hipFree(h->d_items); (void)hipFree(h->d_tile_slots);
    (void)hipFree(h->d_direct); (void)hipFree(h->d_err);
    (void)hipFree(h->d_ua_tiles); (void)hipFree(h->d_ua_items); (void)hipFree(h->d_ua_tile_slots);
    (void)hipFree(h->d_map_sums); (void)hipFree(h->d_map_cnts); (void)hipFree(h->d_map_packed); (void)hipFree(h->d_tw_sums); (void)hipFree(h->d_tw_cnts);
    (void)hipFree(h->d_geom_group); (void)hipFree(h->d_shapes);
    (void)hipFree(h->d_map_rec); (void)hipFree(h->d_ua_runs); (void)hipFree(h->d_ua_run_begin);
    (void)hipFree(h->d_runs); (void)hipFree(h->d_run_begin); (void)hipFree(h->d_items_by_slot);
    (void)hipFree(h->d_dyn_cloud); (void)hipFree(h->d_dyn_heads); (void)hipFree(h->d_dyn_cell_of); (void)hipFree(h->d_dyn_count);
    (void)hipFree(h->d_dyn_rec); (void)hipFree(h->d_dyn_rsn); (void)hipFree(h->d_dyn_normals);
    if (!h->acc_external) (void)hipFree(h->d_acc);
    (void)hipFree(h->d_rep);
    (void)hipFree(h->d_heads); (void)hipFree(h->d_membrane); (void)hipFree(h->d_methyl_begin);
    (void)hipFree(h->d_methyl_atoms); (void)hipFree(h->d_aflags); (void)hipFree(h->d_adist);
    (void)hipFree(h->d_arow); (void)hipFree(h->d_aframes);
    (void)hipFree(h->d_lcell_of); (void)hipFree(h->d_lcell_count); (void)hipFree(h->d_lcell_fill);
    (void)hipFree(h->d_lcell_atoms); (void)hipFree(h->d_ltrig);
    (void)hipFree(h->d_stage_xyz); (void)hipFree(h->d_stage_box);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

uint32_t gorder_hip_n_accumulators(const gorder_hip_handle *h) { return h ? h->plan.n_acc : 0; }
uint32_t gorder_hip_ordermap_dims(const gorder_hip_handle *h, uint32_t *nx, uint32_t *ny) {
    if (nx) *nx = h ? h->map_nx : 0;
    if (ny) *ny = h ? h->map_ny : 0;
    return h ? h->map_nx * h->map_ny : 0;
}

int gorder_hip_set_stream(gorder_hip_handle *h, void *hip_stream) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return GORDER_OK;
}

int gorder_hip_plan(const gorder_hip_handle *h, gorder_hip_plan_t *plan) {
    if (!h || !plan) return GORDER_ERR_INVALID_ARGUMENT;
    plan->n_tiles = (uint32_t)h->plan.tiles.size();
    plan->block_threads = kBlock;
    plan->max_window_atoms = h->plan.max_window;
    plan->n_direct_items = (uint32_t)h->plan.direct.size();
    plan->frames_per_stage = (uint32_t)h->frames_per_stage;
    plan->lds_bytes = (uint32_t)h->lds_bytes;
    return GORDER_OK;
}

// ---- leaflet assignment rows for a batch (host part of leaflets.rs:435-441, 1437-1472) --------
static int run_leaflets(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                        const std::vector<uint32_t> &aframes, uint32_t row0) {
    if (aframes.empty()) return GORDER_OK;
    int st;
    if ((st = ensure(h, &h->d_aframes, &h->aframes_cap, aframes.size())) != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_aframes, aframes.data(), aframes.size() * sizeof(uint32_t),
                              hipMemcpyHostToDevice, h->stream));
    const gorder_leaflets_t &lf = h->tables.leaflets;
    LeafletArgs la{};
    la.xyz = d_xyz; la.box9 = d_box; la.n_atoms = h->plan.n_atoms;
    la.aframes = h->d_aframes; la.row0 = row0; la.aflags = h->d_aflags; la.adist = h->d_adist;
    la.n_mol_total = h->plan.n_mol_total; la.heads = h->d_heads;
    la.membrane = h->d_membrane; la.n_membrane = lf.n_membrane;
    la.methyl_begin = h->d_methyl_begin; la.methyl_atoms = h->d_methyl_atoms;
    la.dim = lf.normal_dim; la.flip = lf.flip ? 1 : 0; la.pbc = h->tables.handle_pbc ? 1 : 0;
    la.err = h->d_err;
    if (lf.method == GORDER_LEAFLETS_GLOBAL) {
        hipLaunchKernelGGL(k_leaflets_global, dim3((uint32_t)aframes.size()), dim3(1024), 0, h->stream, la);
    } else if (lf.method == GORDER_LEAFLETS_INDIVIDUAL) {
        // gridDim.y <= 65535: launch in slabs
        size_t done = 0;
        while (done < aframes.size()) {
            const uint32_t ny = (uint32_t)std::min<size_t>(aframes.size() - done, 65535);
            LeafletArgs lb = la;
            lb.aframes = h->d_aframes + done;
            lb.row0 = row0 + (uint32_t)done;
            // adist is only written by the last slab's last row
            if (done + ny < aframes.size()) lb.adist = nullptr;
            hipLaunchKernelGGL(k_leaflets_individual, dim3((la.n_mol_total + 255) / 256, ny), dim3(256), 0,
                               h->stream, lb);
            done += ny;
        }
    }
    else if (lf.method == GORDER_LEAFLETS_LOCAL) {
        const size_t ncell = (size_t)kLocalMaxCells1D * kLocalMaxCells1D;
        LocalArgs lo{};
        lo.xyz = d_xyz; lo.box9 = d_box; lo.n_atoms = h->plan.n_atoms;
        lo.aflags = h->d_aflags; lo.adist = h->d_adist; lo.n_mol_total = h->plan.n_mol_total;
        lo.heads = h->d_heads; lo.membrane = h->d_membrane; lo.n_membrane = lf.n_membrane;
        lo.dim = lf.normal_dim; lo.flip = lf.flip ? 1 : 0; lo.pbc = h->tables.handle_pbc ? 1 : 0;
        lo.radius = lf.radius;
        lo.radius_thr = local_radius_threshold(lf.radius);
        lo.cell_of = h->d_lcell_of; lo.trig = h->d_ltrig; lo.cell_count = h->d_lcell_count;
        lo.cell_fill = h->d_lcell_fill; lo.rsn = reinterpret_cast<float *>(h->d_lcell_atoms); lo.err = h->d_err;
        for (size_t done = 0; done < aframes.size(); done += kLocalSlab) {
            const uint32_t ns = (uint32_t)std::min<size_t>(aframes.size() - done, kLocalSlab);
            lo.aframes = h->d_aframes + done;
            lo.n_slab = ns;
            lo.row0 = row0 + (uint32_t)done;
            lo.write_dist_frame = (done + ns == aframes.size()) ? (int)ns - 1 : -1;
            HIP_TRY(h, hipMemsetAsync(h->d_lcell_count, 0, ns * (ncell + 1) * sizeof(uint32_t), h->stream));
            HIP_TRY(h, hipMemsetAsync(h->d_lcell_fill, 0, ns * ncell * sizeof(uint32_t), h->stream));
            const dim3 ga((lf.n_membrane + 255) / 256, ns);
            hipLaunchKernelGGL(k_local_bin, ga, dim3(256), 0, h->stream, lo);
            hipLaunchKernelGGL(k_local_scan, dim3(ns), dim3(1024), 0, h->stream, lo);
            hipLaunchKernelGGL(k_local_scatter, ga, dim3(256), 0, h->stream, lo);
            hipLaunchKernelGGL(k_local_flags, dim3((lo.n_mol_total + 3) / 4, ns), dim3(256), 0, h->stream, lo);
        }
    }
    HIP_TRY(h, hipGetLastError());
    return GORDER_OK;
}

int gorder_hip_submit_device(gorder_hip_handle *h, const float *d_xyz, const float *d_box,
                             const uint64_t *frame_index, uint32_t n_frames) {
    if (!h || !d_xyz || !frame_index) return GORDER_ERR_INVALID_ARGUMENT;
    const bool pbc = h->tables.handle_pbc != 0;
    if (pbc && !d_box) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "box required when handle_pbc = 1");
    if (((uintptr_t)d_xyz & 15u) != 0) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "d_xyz must be 16-byte aligned");
    if (n_frames == 0) return GORDER_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const Plan &p = h->plan;
    const gorder_leaflets_t &lf = h->tables.leaflets;
    int st;

    // ---- leaflet assignment rows of this batch; row 0 = assignment carried over from earlier
    // batches (AssignedLeaflets::local, leaflets.rs:1371-1380), rows 1.. = assignment frames here
    const bool leaflets = lf.method != GORDER_LEAFLETS_NONE;
    size_t n_new_rows = 0;
    if (leaflets) {
        std::vector<uint32_t> arow(n_frames), aframes;
        uint32_t cur = 0;
        bool have = h->have_assignment;
        uint64_t last_assign_frame = h->assignment_frame;
        for (uint32_t f = 0; f < n_frames; f++) {
            if (lf.method != GORDER_LEAFLETS_MANUAL && should_assign(lf.frequency, frame_index[f])) {
                aframes.push_back(f);
                cur = (uint32_t)aframes.size();
                have = true;
                last_assign_frame = frame_index[f];
            }
            if (!have) return fail(h, GORDER_ERR_LEAFLETS_NOT_PRIMED, "no leaflet assignment for the first frame");
            arow[f] = cur;
        }
        const size_t rows = aframes.size() + 1;
        if (rows > h->aflags_rows) {
            uint8_t *nb = nullptr;
            const size_t nrows = rows + rows / 4;
            HIP_TRY(h, hipMalloc((void **)&nb, nrows * (size_t)p.n_mol_total));
            if (h->d_aflags) {
                // stream-ordered: the handle's stream is non-blocking, a null-stream copy would not be
                HIP_TRY(h, hipMemcpyAsync(nb, h->d_aflags, p.n_mol_total, hipMemcpyDeviceToDevice, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                HIP_TRY(h, hipFree(h->d_aflags));
            }
            h->d_aflags = nb;
            h->aflags_rows = nrows;
        }
        if ((st = ensure(h, &h->d_arow, &h->arow_cap, n_frames)) != GORDER_OK) return st;
        HIP_TRY(h, hipMemcpyAsync(h->d_arow, arow.data(), n_frames * sizeof(uint32_t), hipMemcpyHostToDevice,
                                  h->stream));
        if ((st = run_leaflets(h, d_xyz, d_box, aframes, 1)) != GORDER_OK) return st;
        h->have_assignment = true;
        h->assignment_frame = last_assign_frame;
        n_new_rows = aframes.size();
    }
    if (pbc) {
        hipLaunchKernelGGL(k_check_box, dim3((n_frames + 255) / 256), dim3(256), 0, h->stream, d_box, n_frames,
                           h->d_err);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->extra.tw && h->n_frames + n_frames > h->tw_cap) {   // grow the per-frame rows (timewise.rs:183-186)
        const size_t row = 3 * (size_t)p.n_acc;
        const uint64_t ncap = (h->n_frames + n_frames) * 2;
        unsigned long long *ns = nullptr, *nc = nullptr;
        HIP_TRY(h, hipMalloc((void **)&ns, ncap * row * sizeof(unsigned long long)));
        HIP_TRY(h, hipMalloc((void **)&nc, ncap * row * sizeof(unsigned long long)));
        // everything on the handle's own (non-blocking) stream: a null-stream memset / copy is NOT ordered
        // with the kernels that follow and left stale rows behind (seen as a wrong error estimate)
        HIP_TRY(h, hipMemsetAsync(ns, 0, ncap * row * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, hipMemsetAsync(nc, 0, ncap * row * sizeof(unsigned long long), h->stream));
        if (h->d_tw_sums) {
            HIP_TRY(h, hipMemcpyAsync(ns, h->d_tw_sums, h->n_frames * row * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(nc, h->d_tw_cnts, h->n_frames * row * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            HIP_TRY(h, hipFree(h->d_tw_sums));
            HIP_TRY(h, hipFree(h->d_tw_cnts));
        }
        h->d_tw_sums = ns; h->d_tw_cnts = nc; h->tw_cap = ncap;
    }
    FrameArgs a{};
    a.xyz = d_xyz; a.box9 = d_box; a.n_atoms = p.n_atoms; a.n_frames = n_frames;
    a.pbc = pbc ? 1 : 0;
    a.nx = h->tables.normal[0]; a.ny = h->tables.normal[1]; a.nz = h->tables.normal[2]; a.n2 = h->n2; a.n2sq = h->n2sq;
    a.leaflets = leaflets ? 1 : 0; a.aflags = h->d_aflags; a.arow = h->d_arow; a.n_mol_total = p.n_mol_total;
    a.acc = h->d_acc; a.rep = h->d_rep; a.n_rep = h->n_rep; a.n_acc = p.n_acc; a.err = h->d_err;
    if ((st = launch_orders(h, a)) != GORDER_OK) return st;
    h->rep_dirty = true;
    if (n_new_rows) {   // newest assignment becomes the carry row of the next batch
        HIP_TRY(h, hipMemcpyAsync(h->d_aflags, h->d_aflags + n_new_rows * (size_t)p.n_mol_total, p.n_mol_total,
                                  hipMemcpyDeviceToDevice, h->stream));
    }
    h->n_frames += n_frames;
    return GORDER_OK;
}

int gorder_hip_submit_host(gorder_hip_handle *h, const float *xyz, const float *box, const uint64_t *frame_index,
                           uint32_t n_frames) {
    if (!h || !xyz || !frame_index) return GORDER_ERR_INVALID_ARGUMENT;
    if (n_frames == 0) return GORDER_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int st;
    const size_t nx = (size_t)n_frames * h->plan.n_atoms * 3u, nb = (size_t)n_frames * 9u;
    // the staging buffer may still be read by kernels of the previous batch
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if ((st = ensure(h, &h->d_stage_xyz, &h->stage_xyz_cap, nx)) != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(h->d_stage_xyz, xyz, nx * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (box) {
        if ((st = ensure(h, &h->d_stage_box, &h->stage_box_cap, nb)) != GORDER_OK) return st;
        HIP_TRY(h, hipMemcpyAsync(h->d_stage_box, box, nb * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    return gorder_hip_submit_device(h, h->d_stage_xyz, box ? h->d_stage_box : nullptr, frame_index, n_frames);
}

int gorder_hip_prime_leaflets(gorder_hip_handle *h, const float *d_xyz, const float *d_box, uint64_t frame_index) {
    if (!h || !d_xyz) return GORDER_ERR_INVALID_ARGUMENT;
    const gorder_leaflets_t &lf = h->tables.leaflets;
    if (lf.method == GORDER_LEAFLETS_NONE || lf.method == GORDER_LEAFLETS_MANUAL) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->aflags_rows < 2) {
        uint8_t *nb = nullptr;
        HIP_TRY(h, hipMalloc((void **)&nb, 2 * (size_t)h->plan.n_mol_total));
        if (h->d_aflags) HIP_TRY(h, hipFree(h->d_aflags));
        h->d_aflags = nb;
        h->aflags_rows = 2;
    }
    std::vector<uint32_t> aframes(1, 0);
    const int st = run_leaflets(h, d_xyz, d_box, aframes, 0);
    if (st != GORDER_OK) return st;
    h->have_assignment = true;
    h->assignment_frame = frame_index;
    return GORDER_OK;
}

int gorder_hip_set_manual_leaflets(gorder_hip_handle *h, const uint8_t *flags, uint64_t frame_index) {
    if (!h || !flags) return GORDER_ERR_INVALID_ARGUMENT;
    if (h->tables.leaflets.method != GORDER_LEAFLETS_MANUAL) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const uint32_t n = h->plan.n_mol_total;
    std::vector<uint8_t> tmp(n);
    for (uint32_t i = 0; i < n; i++) tmp[i] = (uint8_t)((flags[i] & 1) ^ (h->tables.leaflets.flip ? 1 : 0));
    if (h->aflags_rows < 1) {
        HIP_TRY(h, hipMalloc((void **)&h->d_aflags, 2 * (size_t)n));
        h->aflags_rows = 2;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->d_aflags, tmp.data(), n, hipMemcpyHostToDevice));
    h->have_assignment = true;
    h->assignment_frame = frame_index;
    return GORDER_OK;
}

int gorder_hip_synchronize(gorder_hip_handle *h) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_device_error(h);
}

int gorder_hip_finish(gorder_hip_handle *h, int64_t *sums, uint64_t *counts, int64_t *map_sums,
                      uint64_t *map_counts, uint64_t *n_frames_analyzed) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    int st = fold_replicas(h);
    if (st != GORDER_OK) return st;
    if ((st = fold_maps(h)) != GORDER_OK) return st;
    st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const uint32_t n = h->plan.n_acc;
    std::vector<unsigned long long> raw(4 * (size_t)n + 1);
    HIP_TRY(h, hipMemcpy(raw.data(), h->d_acc, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const bool lf = h->tables.leaflets.method != GORDER_LEAFLETS_NONE;
    for (uint32_t s = 0; s < n; s++) {
        const int64_t tot = (int64_t)raw[s], up = (int64_t)raw[n + s];
        const uint64_t ctot = raw[2 * (size_t)n + s], cup = raw[3 * (size_t)n + s];
        if (sums) {
            sums[s] = tot;
            sums[n + s] = lf ? up : 0;
            sums[2 * (size_t)n + s] = lf ? tot - up : 0;   // every sample is upper or lower (bond.rs:199-213)
        }
        if (counts) {
            counts[s] = ctot;
            counts[n + s] = lf ? cup : 0;
            counts[2 * (size_t)n + s] = lf ? ctot - cup : 0;
        }
    }
    if (n_frames_analyzed) *n_frames_analyzed = raw[4 * (size_t)n];
    if (h->extra.maps) {
        const size_t nmap = 3 * (size_t)n * h->map_nx * h->map_ny;
        if (map_sums) HIP_TRY(h, hipMemcpy(map_sums, h->d_map_sums, nmap * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (map_counts) HIP_TRY(h, hipMemcpy(map_counts, h->d_map_cnts, nmap * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return GORDER_OK;
}

int gorder_hip_normals(gorder_hip_handle *h, float *normals, uint32_t *n_points) {
    if (!h || !h->dyn) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const uint32_t n_mol = h->plan.n_mol_total;
    if (h->last_normals.size() < 4 * (size_t)n_mol) return fail(h, GORDER_ERR_INVALID_ARGUMENT, "no frame submitted yet");
    for (uint32_t m = 0; m < n_mol; m++) {
        if (normals) for (int d = 0; d < 3; d++) normals[3 * (size_t)m + d] = h->last_normals[4 * (size_t)m + d];
        if (n_points) n_points[m] = (uint32_t)h->last_normals[4 * (size_t)m + 3];
    }
    return GORDER_OK;
}

int gorder_hip_timewise(gorder_hip_handle *h, int64_t *tw_sums, uint64_t *tw_counts, uint64_t capacity_frames) {
    if (!h || !h->extra.tw || !tw_sums || !tw_counts) return GORDER_ERR_INVALID_ARGUMENT;
    if (capacity_frames < h->n_frames) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    const size_t n = h->n_frames * 3 * (size_t)h->plan.n_acc;
    if (n) {
        HIP_TRY(h, hipMemcpy(tw_sums, h->d_tw_sums, n * sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(tw_counts, h->d_tw_cnts, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return GORDER_OK;
}

int gorder_hip_leaflets(gorder_hip_handle *h, uint8_t *flags, uint64_t *assignment_frame) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    if (!h->have_assignment) return GORDER_ERR_LEAFLETS_NOT_PRIMED;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    if (flags) HIP_TRY(h, hipMemcpy(flags, h->d_aflags, h->plan.n_mol_total, hipMemcpyDeviceToHost));
    if (assignment_frame) *assignment_frame = h->assignment_frame;
    return GORDER_OK;
}

int gorder_hip_leaflet_distances(gorder_hip_handle *h, float *dist) {
    if (!h || !dist || !h->d_adist) return GORDER_ERR_INVALID_ARGUMENT;
    const int st = gorder_hip_synchronize(h);
    if (st != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpy(dist, h->d_adist, sizeof(float) * h->plan.n_mol_total, hipMemcpyDeviceToHost));
    return GORDER_OK;
}

int gorder_hip_accumulators_device(gorder_hip_handle *h, void **d_ptr, uint64_t *n_u64) {
    if (!h || !d_ptr || !n_u64) return GORDER_ERR_INVALID_ARGUMENT;
    {   // the packed block must be complete before a collective reads it (stream-ordered)
        const int st = fold_replicas(h);
        if (st != GORDER_OK) return st;
    }
    *d_ptr = h->d_acc;
    *n_u64 = h->acc_words;
    return GORDER_OK;
}

int gorder_hip_export_maps(gorder_hip_handle *h, void *d_sums, void *d_counts, uint64_t n_u64) {
    if (!h || !h->extra.maps || !d_sums || !d_counts) return GORDER_ERR_INVALID_ARGUMENT;
    const uint64_t n = 3ull * h->plan.n_acc * h->map_nx * h->map_ny;
    if (n_u64 < n) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    const int st = fold_maps(h);
    if (st != GORDER_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(d_sums, h->d_map_sums, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(d_counts, h->d_map_cnts, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return check_device_error(h);
}

int gorder_hip_bind_accumulators(gorder_hip_handle *h, void *d_ptr, uint64_t n_u64) {
    if (!h || !d_ptr || n_u64 < h->acc_words || ((uintptr_t)d_ptr & 7u)) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const int st = fold_replicas(h);
        if (st != GORDER_OK) return st;
    }
    HIP_TRY(h, hipMemcpyAsync(d_ptr, h->d_acc, h->acc_words * sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (!h->acc_external) HIP_TRY(h, hipFree(h->d_acc));
    h->d_acc = (unsigned long long *)d_ptr;
    h->acc_external = true;
    return GORDER_OK;
}

uint64_t gorder_hip_last_error_index(const gorder_hip_handle *h) { return h ? h->err_index : 0; }
const char *gorder_hip_last_error_message(const gorder_hip_handle *h) { return h ? h->err_msg.c_str() : ""; }

int gorder_hip_kernel_time(gorder_hip_handle *h, double *ms, uint64_t *launches, int reset) {
    if (!h) return GORDER_ERR_INVALID_ARGUMENT;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (auto &ev : h->timing) {
        float t = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&t, ev.first, ev.second));
        h->timing_ms += t;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->timing.clear();
    if (ms) *ms = h->timing_ms;
    if (launches) *launches = h->timing_launches;
    if (reset) { h->timing_ms = 0.0; h->timing_launches = 0; }
    return GORDER_OK;
}

}  // extern "C"
