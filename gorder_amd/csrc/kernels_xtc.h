// kernels_xtc.h — k_xtc_decode: GROMACS xdr3dcoord bit streams -> f32 coordinates of the analysed atoms, on the device.
// Part of the single translation unit gorder_hip.hip; device code for gfx950 only.
//
// What it replaces: the decoding half of the reference's reader (groan_rs GroupXtcReader -> molly 0.5.0, reached from
// common.rs:283-304) — "the reference's true bottleneck" (SURVEY §6, §8f row 1).  The host only copies the compressed
// blocks (gorder_xtc_pack_window); the bit stream of a frame is strictly sequential (every field's position and the
// adaptive `smallidx` depend on everything before it), so the parallel axis is the FRAME: one lane decodes one frame,
// a wave 64 frames, a window of thousands of frames a few hundred waves.  Same integers, same `int * (1 / precision)`
// as gorder_xtc_next (xtc_reader.cpp): the two decoders are compared bit for bit in tests/test_xtc_device_gpu.py.
//
// Shape of the loop.  The reference algorithm is "one full-width atom, then a run of 0..8 small atoms"; here every
// iteration of every lane decodes exactly ONE atom — full-width or small, by the lane's own state — so the atom
// counter is wave-uniform (the slot lookups are scalar loads, the stores of a wave go to the same atom of 64 frames)
// and the trip count is the number of atoms up to the last analysed one, not a data-dependent quantity.
// The bit reader keeps 128 bits of the stream in registers and the next 64 prefetched from the lane's ring in LDS, which
// is refilled in 64-byte pieces at the service points of the loop (see k_xtc_decode): no load is on the dependency
// chain of a field, and no instruction of the loop waits for memory.
#pragma once

#include "../../include/gorder_xtc.h"

namespace {

constexpr int kXtcFirstIdx = 9, kXtcLastIdx = 73;
constexpr uint32_t kXtcGroup = 8;      // atoms written out together (see k_xtc_decode)
constexpr uint32_t kXtcWaves = 1;      // waves per workgroup (co-resident waves do not shorten a wave; 12 per CU measured 25 % slower)
constexpr uint32_t kXtcRingWords = 32;  // a lane's input ring in LDS: 256 bytes of its stream, refilled 64 bytes at a time
constexpr uint32_t kXtcRingPitch = 33;  // words per lane (bank spread)
constexpr uint32_t kXtcPitch = 65;     // floats per (atom, coordinate) row of the LDS staging: 64 frames + 1 (bank spread)
__device__ const uint32_t kXtcMagic[kXtcLastIdx] = {
    0,       0,       0,       0,       0,       0,       0,       0,       0,       8,        10,       12,
    16,      20,      25,      32,      40,      50,      64,      80,      101,     128,      161,      203,
    256,     322,     406,     512,     645,     812,     1024,    1290,    1625,    2048,     2580,     3250,
    4096,    5060,    6501,    8192,    10321,   13003,   16384,   20642,   26007,   32768,    41285,    52015,
    65536,   82570,   104031,  131072,  165140,  208063,  262144,  330280,  416127,  524287,   660561,   832255,
    1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304, 5284491, 6658042, 8388607,  10568983, 13316085,
    16777216};

struct XtcBits {
    const unsigned long long *ring;        // this lane's ring of kXtcRingWords stream words in LDS (k_xtc_decode keeps it filled)
    uint32_t rd;                           // words of the stream handed to the window so far
    unsigned long long w0, w1, pre;        // w0:w1 = the next 128 bits of the stream, MSB first; pre = the raw word behind them
    uint32_t off;                          // bits of w0 already taken
    unsigned long long taken;              // bits taken in all
    __device__ __forceinline__ void open(const unsigned long long *lane_ring) {   // the first 3 words are in the ring
        ring = lane_ring;
        w0 = __builtin_bswap64(ring[0]);
        w1 = __builtin_bswap64(ring[1]);
        pre = ring[2];
        rd = 3;
        off = 0;
        taken = 0;
    }
    __device__ __forceinline__ unsigned long long peek() const {        // the next 64 bits
        return off ? (w0 << off) | (w1 >> (64u - off)) : w0;
    }
    __device__ __forceinline__ unsigned long long take(uint32_t n) {   // 0 <= n <= 64
        const unsigned long long v = n ? peek() >> (64u - n) : 0ull;
        skip(n);
        return v;
    }
    __device__ __forceinline__ void skip(uint32_t n) {                  // 0 <= n <= 64
        off += n;
        taken += n;
        if (off >= 64u) {
            off -= 64u;
            w0 = w1;
            w1 = __builtin_bswap64(pre);
            pre = ring[rd & (kXtcRingWords - 1u)];      // an LDS read, waited for at the next refill only
            rd++;
        }
    }
};

// a reader over ONE peeked word: the fields of an atom whose bits all lie within the next 64 (the usual case) are cut
// out of a register, and the stream moves once per atom
struct XtcPeek {
    unsigned long long p;
    uint32_t used;
    __device__ __forceinline__ unsigned long long take(uint32_t n) {   // used + n <= 64
        const unsigned long long v = n ? (p << used) >> (64u - n) : 0ull;
        used += n;
        return v;
    }
};

// v / s by the precomputed floor(2^64 / s): the estimate is at most two short (xtc_reader.cpp BitReader::ints)
__device__ __forceinline__ unsigned long long xtc_div(unsigned long long v, uint32_t s, unsigned long long recip, uint32_t &rem) {
    unsigned long long q = __umul64hi(v, recip);
    unsigned long long r = v - q * s;
    if (r >= s) { r -= s; q++; }
    if (r >= s) { r -= s; q++; }
    rem = (uint32_t)r;
    return q;
}

// Three integers packed as one mixed-radix number of nbits (1..72) bits with radices (-, s1, s2).  The number is
// stored in chunks of 8 bits, first chunk least significant, the last (partial) chunk on top.
__device__ __forceinline__ void xtc_ints(XtcBits &b, uint32_t nbits, uint32_t s1, uint32_t s2, unsigned long long r1,
                                         unsigned long long r2, int (&out)[3]) {
    uint32_t c1, c2;
    if (nbits <= 32u) {
        // the usual small-offset atom (and full atoms of tiny boxes): everything in 32 bits — a 64 x 64 -> 128
        // multiplication is seven quarter-rate integer multiplications on this hardware, this path has four in all.
        // floor(2^32 / s) = floor(2^64 / s) >> 32; the estimate is at most one short (two repairs as above).
        const uint32_t m = (nbits - 1u) >> 3, rem = nbits - 8u * m;
        const uint32_t x = (uint32_t)b.take(nbits);
        uint32_t v = (x & ((1u << rem) - 1u)) << (8u * m);
        if (m) v |= __builtin_bswap32((x >> rem) << (32u - 8u * m));
        auto div32 = [](uint32_t n, uint32_t s, uint32_t recip, uint32_t &r) {
            uint32_t q = __umulhi(n, recip);
            r = n - q * s;
            if (r >= s) { r -= s; q++; }
            if (r >= s) { r -= s; q++; }
            return q;
        };
        const uint32_t q2 = div32(v, s2, (uint32_t)(r2 >> 32), c2);
        out[0] = (int)div32(q2, s1, (uint32_t)(r1 >> 32), c1);
    } else if (nbits <= 64u) {
        const uint32_t m = (nbits - 1u) >> 3, rem = nbits - 8u * m;     // m full chunks, then rem (1..8) bits
        const unsigned long long x = b.take(nbits);
        unsigned long long v = (x & ((1ull << rem) - 1ull)) << (8u * m);
        if (m) v |= __builtin_bswap64((x >> rem) << (64u - 8u * m));
        const unsigned long long q2 = xtc_div(v, s2, r2, c2);
        const unsigned long long q1 = xtc_div(q2, s1, r1, c1);
        out[0] = (int)(uint32_t)q1;
    } else {
        // 65..72 bits (boxes beyond ~2 million grid steps per edge): long division over 32-bit limbs
        uint32_t l0 = __builtin_bswap32((uint32_t)b.take(32)), l1 = __builtin_bswap32((uint32_t)b.take(32));
        uint32_t l2 = (uint32_t)b.take(nbits - 64u);
        auto div96 = [&](uint32_t s, unsigned long long recip) {
            uint32_t r;
            l2 = (uint32_t)xtc_div(l2, s, recip, r);
            l1 = (uint32_t)xtc_div(((unsigned long long)r << 32) | l1, s, recip, r);
            l0 = (uint32_t)xtc_div(((unsigned long long)r << 32) | l0, s, recip, r);
            return r;
        };
        c2 = div96(s2, r2);
        c1 = div96(s1, r1);
        out[0] = (int)l0;
    }
    out[1] = (int)c1;
    out[2] = (int)c2;
}

// grid = ceil(n_frames / 64) blocks of ONE wave; lane = frame.
//   blob, frames : what gorder_xtc_pack_window produced (device copies)
//   natoms       : atoms per frame in the file;  n_stop: atoms to go through (up to the last analysed one)
//   slot_of      : [natoms] output slot of an atom or -1 (null: every atom, slot = atom)
//   out          : [n_frames][n_out][3]
//   stat, short_list : null, or 2 words (zeroed by the caller) + [n_frames] words: see the end of the kernel
__global__ __launch_bounds__(64 * kXtcWaves) void k_xtc_decode(const uint8_t *__restrict__ blob, unsigned long long blob_bytes,
                                                   const gorder_xtc_frame_t *__restrict__ frames, uint32_t n_frames,
                                                   uint32_t natoms, const int32_t *__restrict__ slot_of, uint32_t n_stop,
                                                   float *__restrict__ out, uint32_t n_out, uint32_t *err,
                                                   uint32_t *stat, uint32_t *short_list) {
    __shared__ uint32_t l_magic[kXtcLastIdx];
    __shared__ unsigned long long l_recip[kXtcLastIdx];
    __shared__ double l_inv[kXtcLastIdx];
    for (uint32_t k = threadIdx.x; k < (uint32_t)kXtcLastIdx; k += 64u * kXtcWaves) {
        const uint32_t m = kXtcMagic[k];
        l_magic[k] = m;
        // floor(2^64 / m): a power of two divides 2^64 exactly, for anything else it is floor((2^64 - 1) / m)
        l_recip[k] = m == 0u ? 0ull : ((m & (m - 1u)) == 0u ? 1ull << (64 - __builtin_ctz(m)) : ~0ull / m);
        l_inv[k] = m == 0u ? 0.0 : 1.0 / (double)m;
    }
    __syncthreads();
    // a wave = 64 frames; the waves of a workgroup share nothing but the tables above and a CU: with a few thousand
    // frames per launch there are far fewer waves than SIMDs, and the dispatcher would give each a SIMD of its own,
    // where every dependent instruction, LDS access and load is waited for in full (measured: 190 instructions per
    // atom take 2 100 cycles).  Twelve waves per workgroup put three on each SIMD of one CU instead.
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t frame0 = (blockIdx.x * kXtcWaves + wave) * 64u;
    if (frame0 >= n_frames) return;              // (no workgroup barrier below this line)
    const uint32_t fr = frame0 + lane;
    const bool live = fr < n_frames;
    const gorder_xtc_frame_t d = frames[live ? fr : n_frames - 1u];
    float *o = out + (size_t)(live ? fr : 0u) * n_out * 3u;      // (the raw path below)
    const float inv_p = d.inv_precision;
    // the frame's region of the blob: the block rounded up to whole 64-byte pieces plus one piece of zeros
    const unsigned long long region = (((unsigned long long)d.n_bytes + 63ull) & ~63ull) + 64ull;
    bool bad = (d.offset & 63ull) != 0ull || d.offset + region > blob_bytes;
    const uint8_t *p = blob + (bad ? 0ull : d.offset);
    const unsigned long long last_piece = bad ? 0ull : region - 64ull;      // (a bad frame reads the blob's first piece, harmlessly)

    // Decoded atoms go through LDS: a lane writing its frame's atom straight to memory is one 4-byte piece in each of
    // 64 cache lines per store instruction.  Instead a lane parks atom i at l_atoms[i mod 16] and, once 8 consecutive
    // atoms are final, the wave writes them frame by frame: 24 lanes = 8 atoms x 3 coordinates = one contiguous 96-byte
    // piece of a frame when the analysed atoms are consecutive in the file (any other order is still correct, only
    // less coalesced), two frames per store instruction.
    __shared__ float l_atoms_all[kXtcWaves * 2u * kXtcGroup * 3u * kXtcPitch];
    float *l_atoms = l_atoms_all + wave * (2u * kXtcGroup * 3u * kXtcPitch);
    const uint32_t n_live = min(64u, n_frames - frame0);
    auto put = [&](const int (&c)[3], uint32_t idx) {        // idx is wave-uniform
        float *q = l_atoms + (idx & (2u * kXtcGroup - 1u)) * 3u * kXtcPitch + lane;
        q[0] = (float)c[0] * inv_p;
        q[kXtcPitch] = (float)c[1] * inv_p;
        q[2u * kXtcPitch] = (float)c[2] * inv_p;
    };
    // the output slot of the (atom, coordinate) this lane writes in the flush of the group that starts at first_atom
    const uint32_t fl_half = lane >= 3u * kXtcGroup ? 1u : 0u, fl_l = lane - fl_half * 3u * kXtcGroup;
    const uint32_t fl_j = fl_l / 3u, fl_c = fl_l - 3u * fl_j;
    auto slot_for = [&](uint32_t first_atom) -> int32_t {
        const uint32_t idx = first_atom + fl_j;
        if (lane >= 6u * kXtcGroup || idx >= n_stop) return -1;
        return slot_of ? slot_of[idx] : (int32_t)idx;
    };
    auto flush = [&](uint32_t first_atom, int32_t slot) {    // atoms first_atom .. first_atom + 7, all final
        // the wave reads what its own lanes wrote: LDS instructions of one wave execute in order, the fences only keep
        // the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lanes 0..23: (atom, coordinate) of the even frames, lanes 24..47: of the odd frames — 96 contiguous bytes each
        const uint32_t half = fl_half, c = fl_c, idx = first_atom + fl_j;
        if (slot >= 0 && (uint32_t)slot < n_out) {
            const float *q = l_atoms + ((idx & (2u * kXtcGroup - 1u)) * 3u + c) * kXtcPitch;
            float *dst = out + ((size_t)frame0 * n_out + (uint32_t)slot) * 3u + c;
#pragma unroll 4
            for (uint32_t f = half; f < n_live; f += 2u) dst[(size_t)f * n_out * 3u] = q[f];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    if (natoms <= 9u) {      // uncompressed small systems: big-endian floats
        for (uint32_t t = 0; t < n_stop; t++) {
            const int32_t slot = slot_of ? slot_of[t] : (int32_t)t;
            if (slot < 0 || (uint32_t)slot >= n_out || !live || bad || d.n_bytes < 12u * natoms) continue;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(p) + 3u * t;
            for (int c = 0; c < 3; c++) o[3u * (size_t)slot + c] = __uint_as_float(__builtin_bswap32(w[c]));
        }
        if (bad && live) raise_error(err, GORDER_ERR_TRAJECTORY_FORMAT, fr, kStageBox);
        return;
    }

    // ---- the input side: each lane's stream goes through a 256-byte ring in LDS, refilled in 64-byte pieces at the
    // same points where the output is flushed.  A piece is REQUESTED at one service point (global loads into registers)
    // and LANDED in the ring at the next one, eight atoms later: no instruction of the loop ever waits for memory, and
    // the wait for the loads no longer waits for the flush's stores either (gfx9 counts both in vmcnt) — they are all
    // one service point old by then.  A lane uses at most 78 bytes between two service points (8 atoms of 72 + 6 bits)
    // and gets up to 128, a corrupt stream just runs into the region's zero piece (the piece ADDRESS is clamped).
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned long long l_in_all[kXtcWaves * 64u * kXtcRingPitch];
    unsigned long long *ring = l_in_all + (wave * 64u + lane) * kXtcRingPitch;
    uint32_t wr = 0;                 // bytes of the stream landed in the ring (a multiple of 64)
    uint32_t n_req = 0;              // pieces requested at the last service point (0..2)
    u32x4 pc[8];
    auto piece_at = [&](uint32_t off) {
        return reinterpret_cast<const u32x4 *>(p + (off < last_piece ? (unsigned long long)off : last_piece));
    };
    auto land = [&](const u32x4 *v) {           // one piece into the ring at stream offset wr (8-byte stores: odd pitch)
        unsigned long long *dst = ring + ((wr >> 3) & (kXtcRingWords - 1u));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dst[2 * k] = (unsigned long long)v[k].x | ((unsigned long long)v[k].y << 32);
            dst[2 * k + 1] = (unsigned long long)v[k].z | ((unsigned long long)v[k].w << 32);
        }
        wr += 64u;
    };
    {   // the first four pieces, straight in
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const u32x4 *src = piece_at(64u * k);
            pc[0] = src[0]; pc[1] = src[1]; pc[2] = src[2]; pc[3] = src[3];
            land(pc);
        }
    }
    XtcBits b;
    b.open(ring);
    // A service point, every kXtcGroup atoms: land what was requested at the last one, flush the finished atoms, request
    // the next pieces and the output slots of the next flush.  Every global load is consumed one service point after it
    // was issued, so the only waits on memory are for operations eight atoms old.
    int32_t slot_next = slot_for(0);
    auto service = [&](uint32_t first_atom) {
        if (n_req >= 1u) land(pc);
        if (n_req == 2u) land(pc + 4);
        flush(first_atom, slot_next);
        slot_next = slot_for(first_atom + kXtcGroup);
        const uint32_t room = 8u * kXtcRingWords - (wr - 8u * b.rd);        // bytes of the ring not holding unread stream
        n_req = min(2u, room / 64u);
        if (n_req >= 1u) { const u32x4 *src = piece_at(wr); pc[0] = src[0]; pc[1] = src[1]; pc[2] = src[2]; pc[3] = src[3]; }
        if (n_req == 2u) { const u32x4 *src = piece_at(wr + 64u); pc[4] = src[0]; pc[5] = src[1]; pc[6] = src[2]; pc[7] = src[3]; }
    };
    int smallidx = d.smallidx;
    if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) { bad = true; smallidx = kXtcFirstIdx; }
    uint32_t sizesmall = l_magic[smallidx];
    unsigned long long rsmall = l_recip[smallidx];
    double sinv = l_inv[smallidx];
    int smallnum = (int)(sizesmall / 2u);
    // The table step after an atom group.  The reference decoder carries two values, `smallnum` and `smaller`, but
    // smallnum == magic[smallidx] / 2 holds at every point (xtc_reader.cpp decode_ints keeps `smaller` ==
    // magic[smallidx - 1] / 2 only to assign it on the way down): one table row, fetched in one go.
    auto adapt = [&](int is_smaller) {
        if (is_smaller == 0) return;
        smallidx += is_smaller;
        if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) { bad = true; smallidx = kXtcFirstIdx; }
        sizesmall = l_magic[smallidx];
        rsmall = l_recip[smallidx];
        sinv = l_inv[smallidx];
        smallnum = (int)(sizesmall / 2u);
    };
    int run = 0, pend = 0;
    uint32_t run_left = 0;          // small atoms of the current run still to come
    bool first = false;             // the next small atom is the first of its run (it swaps places with its predecessor)
    int prev[3] = {0, 0, 0};
    // an atom is final one iteration after it was read (the swap): one iteration past the last analysed atom
    const uint32_t n_iter = min(natoms, n_stop + 1u);
    uint32_t flushed = 0;           // atoms written out so far (a multiple of kXtcGroup)
    // ---- the usual case, for the whole wave: every frame's full-width atom is ONE number of at most 53 bits (box edges
    // below ~100 000 grid steps).  A full-width atom and a small-offset atom are then the SAME computation with other
    // parameters — a mixed-radix number of nb bits with radices (s1, s2), cut out of the next 64 bits of the stream —
    // so the lanes of a wave, which are in the two states at the same time, run ONE copy of it instead of both
    // (measured 2 000 cycles per iteration with two copies).  The two divisions are done in f64: v < 2^53 is exact as a
    // double, floor(v * (1 / s)) is within 1 of the quotient, the remainder by fma is exact and repairs it — 8 f64
    // operations per division where floor(2^64 / s) arithmetic needs 7 quarter-rate integer multiplications.
    const bool simple = d.bitsize >= 1u && d.bitsize <= 53u;
    if (__ballot(!simple) == 0ull) {
        const double ls1 = (double)d.sizeint[1], ls2 = (double)d.sizeint[2];
        const double li1 = 1.0 / ls1, li2 = 1.0 / ls2;
        for (uint32_t t = 0; t < n_iter; t++) {
            if (t == flushed + kXtcGroup + 1u) {     // atoms < t - 1 are final
                service(flushed);
                flushed += kXtcGroup;
            }
            const bool large = run_left == 0u;
            const uint32_t nb = large ? d.bitsize : (uint32_t)smallidx;
            const unsigned long long P = b.peek();
            uint32_t used = nb;
            int c3[3];
            if (nb <= 53u) {
                const double s1 = large ? ls1 : (double)sizesmall, s2 = large ? ls2 : (double)sizesmall;
                const double i1 = large ? li1 : sinv, i2 = large ? li2 : sinv;
                const uint32_t m = (nb - 1u) >> 3, rem = nb - 8u * m;      // m whole chunks (first lowest), rem bits on top
                const unsigned long long x = P >> (64u - nb);
                unsigned long long v = (x & ((1ull << rem) - 1ull)) << (8u * m);
                if (m) v |= __builtin_bswap64((x >> rem) << (64u - 8u * m));
                const double vd = (double)v;
                double q2 = __builtin_floor(vd * i2);
                double r2 = __builtin_fma(-q2, s2, vd);
                if (r2 < 0.0) { q2 -= 1.0; r2 += s2; } else if (r2 >= s2) { q2 += 1.0; r2 -= s2; }
                double q1 = __builtin_floor(q2 * i1);
                double r1 = __builtin_fma(-q1, s1, q2);
                if (r1 < 0.0) { q1 -= 1.0; r1 += s1; } else if (r1 >= s1) { q1 += 1.0; r1 -= s1; }
                c3[0] = (int)(uint32_t)q1;
                c3[1] = (int)(uint32_t)r1;
                c3[2] = (int)(uint32_t)r2;
            } else {      // small offsets beyond 2^17 grid steps (never seen; valid): the general unpacking, from the stream
                xtc_ints(b, nb, sizesmall, sizesmall, rsmall, rsmall, c3);
                used = 0;
            }
            if (large) {
                c3[0] += d.minint[0]; c3[1] += d.minint[1]; c3[2] += d.minint[2];
                int is_smaller = 0;
                used += 1u;
                if ((P >> (63u - nb)) & 1ull) {
                    run = (int)((P >> (58u - nb)) & 31ull);
                    used += 5u;
                    is_smaller = run % 3;
                    run -= is_smaller;
                    is_smaller--;
                }
                b.skip(used);
                prev[0] = c3[0]; prev[1] = c3[1]; prev[2] = c3[2];
                if (run > 0) {
                    run_left = (uint32_t)run / 3u;
                    first = true;
                    pend = is_smaller;
                    if (t + 1u + run_left > natoms) { bad = true; run_left = 0; }
                } else {
                    put(c3, t);
                    adapt(is_smaller);
                }
            } else {
                b.skip(used);
                int cur[3];
                cur[0] = c3[0] + prev[0] - smallnum;
                cur[1] = c3[1] + prev[1] - smallnum;
                cur[2] = c3[2] + prev[2] - smallnum;
                if (first) {             // stored AFTER the second atom of the run (water: O after H)
                    put(cur, t - 1u);
                    put(prev, t);
                    first = false;
                } else {
                    put(cur, t);
                }
                prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
                if (--run_left == 0u) adapt(pend);
            }
        }
    } else
    // ---- every form of the format (edges beyond 2^24 grid steps, numbers of 54..72 bits): field by field
    for (uint32_t t = 0; t < n_iter; t++) {
        if (t == flushed + kXtcGroup + 1u) {     // atoms < t - 1 are final
            service(flushed);
            flushed += kXtcGroup;
        }
        if (run_left == 0u) {
            int cur[3];
            if (d.bitsize == 0u) {
                cur[0] = (int)(uint32_t)b.take(d.bitsizeint & 0xffu);
                cur[1] = (int)(uint32_t)b.take((d.bitsizeint >> 8) & 0xffu);
                cur[2] = (int)(uint32_t)b.take((d.bitsizeint >> 16) & 0xffu);
            } else {
                xtc_ints(b, d.bitsize, d.sizeint[1], d.sizeint[2], d.recip1, d.recip2, cur);
            }
            cur[0] += d.minint[0]; cur[1] += d.minint[1]; cur[2] += d.minint[2];
            int is_smaller = 0;
            if (b.take(1)) {
                run = (int)b.take(5);
                is_smaller = run % 3;
                run -= is_smaller;
                is_smaller--;
            }
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
            if (run > 0) {
                run_left = (uint32_t)run / 3u;
                first = true;
                pend = is_smaller;
                if (t + 1u + run_left > natoms) { bad = true; run_left = 0; }
            } else {
                put(cur, t);
                adapt(is_smaller);
            }
        } else {
            int dd[3], cur[3];
            xtc_ints(b, (uint32_t)smallidx, sizesmall, sizesmall, rsmall, rsmall, dd);
            cur[0] = dd[0] + prev[0] - smallnum;
            cur[1] = dd[1] + prev[1] - smallnum;
            cur[2] = dd[2] + prev[2] - smallnum;
            if (first) {             // stored AFTER the second atom of the run (water: O after H)
                put(cur, t - 1u);
                put(prev, t);
                first = false;
            } else {
                put(cur, t);
            }
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
            if (--run_left == 0u) adapt(pend);
        }
    }
    for (; flushed < n_stop; flushed += kXtcGroup) flush(flushed, slot_for(flushed));
    // Read into the padding?  A frame of which only a leading part was copied (bit 1 of kind) is then SHORT, not corrupt
    // (whatever else went wrong while it decoded zeros): it goes on the caller's list and is decoded elsewhere.
    // stat[0] = short frames, stat[1] = max over frames of bytes needed / bytes given, in units of 2^-16.
    const bool over = b.taken > 8ull * d.n_bytes;
    if (live) {
        if (stat) {
            const unsigned long long q16 = (((b.taken + 7ull) >> 3) << 16) / (d.n_bytes ? d.n_bytes : 1u);
            atomicMax(&stat[1], (uint32_t)(q16 > 0xffffffffull ? 0xffffffffull : q16));
        }
        if (over && (d.kind & 2u) && stat && short_list) short_list[atomicAdd(&stat[0], 1u)] = fr;
        else if (bad || over) raise_error(err, GORDER_ERR_TRAJECTORY_FORMAT, fr, kStageBox);
    }
}

}  // namespace
