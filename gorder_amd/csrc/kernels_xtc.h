// kernels_xtc.h — k_xtc_scan + k_xtc_chunks: GROMACS xdr3dcoord bit streams -> f32 coordinates of the analysed atoms, on
// the device.  Part of the single translation unit gorder_hip.hip; device code for gfx950 only.
//
// What it replaces: the decoding half of the reference's reader (groan_rs GroupXtcReader -> molly 0.5.0, reached from
// common.rs:283-304) — "the reference's true bottleneck" (SURVEY §6, §8f row 1).  The host only copies the compressed
// blocks (gorder_xtc_pack_window).  Same integers, same `int * (1 / precision)` as gorder_xtc_next (xtc_reader.cpp): the
// two decoders are compared bit for bit in tests/test_xtc_device_gpu.py.
//
// The bit stream of a frame is sequential — every field's position and the adaptive `smallidx` depend on everything
// before it — but WHERE the fields lie can be found without decoding them: a group is one full-width atom of a width
// the frame header fixes, one flag bit, and with the flag five bits that give the number of small atoms behind it
// (each `smallidx` bits wide) and the step of `smallidx`.  So the work is cut in two:
//   k_xtc_scan   (one wave per frame): walks the groups, reading six bits per group — 64 groups at a time where the
//                stream allows —, and writes a checkpoint — bit offset, smallidx, run length, atom index — at the
//                first group boundary at or behind every kXtcChunk-th atom;
//   k_xtc_chunks (one lane per (frame, chunk)): decodes a chunk of ~kXtcChunk atoms from its checkpoint; a group
//                starts with an absolute atom, so a chunk needs nothing from the atoms before it.
// A batch of 3 566 frames of 25 088 atoms is 3 566 waves in the first kernel (a short one: no divisions, no output)
// and 5 460 waves in the second, where the first version of this file (one lane per frame for the whole decode) kept
// 6 % of the machine busy for 14 ms whatever the batch size.
//
// Shape of the chunk loop.  The reference algorithm is "one full-width atom, then a run of 0..8 small atoms"; here
// every iteration of every lane decodes exactly ONE atom — full-width or small, by the lane's own state — so the atom
// counter (relative to the chunk's first atom) is wave-uniform.  The bit reader keeps 128 bits of the stream in
// registers and the next 64 prefetched from the lane's ring in LDS, which is refilled in 64-byte pieces at the service
// points of the loop: no load is on the dependency chain of a field, and no instruction of the loop waits for memory.
#pragma once

#include "../../include/gorder_xtc.h"

namespace {

constexpr int kXtcFirstIdx = 9, kXtcLastIdx = 73;
constexpr uint32_t kXtcGroup = 8;      // atoms written out together (see k_xtc_chunks)
constexpr uint32_t kXtcChunk = 256;    // atoms between two checkpoints (a chunk starts at the first group boundary behind)
// where a chunk starts in its frame's stream, and the decoder's state there: `state` = smallidx | run << 8 (the run
// length is carried from group to group: a group without the flag repeats the last one's)
struct XtcCheckpoint { unsigned long long bit; uint32_t atom; uint32_t state; };
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
    const unsigned long long *ring;        // this lane's ring of kXtcRingWords stream words in LDS (k_xtc_chunks keeps it filled)
    uint32_t rd;                           // words of the stream handed to the window so far
    unsigned long long w0, w1, pre;        // w0:w1 = the next 128 bits of the stream, MSB first; pre = the raw word behind them
    uint32_t off;                          // bits of w0 already taken
    unsigned long long taken;              // bits taken in all
    // the stream from bit `start` (< 512) of what the ring holds from its word 0 on; words start/64 .. +2 are in the ring
    __device__ __forceinline__ void open(const unsigned long long *lane_ring, uint32_t start = 0) {
        ring = lane_ring;
        const uint32_t w = start >> 6;
        w0 = __builtin_bswap64(ring[w]);
        w1 = __builtin_bswap64(ring[w + 1u]);
        pre = ring[w + 2u];
        rd = w + 3u;
        off = start & 63u;
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

// ---- pass 1: where the chunks start ---------------------------------------------------------------------------------
// block = 256 threads = 4 waves = 4 frames, a WAVE per frame; grid = ceil(n_frames / 4).
//   blob, frames : what gorder_xtc_pack_window produced (device copies)
//   natoms       : atoms per frame in the file;  n_stop: atoms to go through (up to the last analysed one)
//   cp           : [n_frames][n_chunks + 1] checkpoints, n_chunks = ceil(n_stop / kXtcChunk); entry c = the first group
//                  boundary with at least c * kXtcChunk atoms before it, the last entry = where the walk ended
//   slot_of, out : used here only for the uncompressed frames of files with <= 9 atoms (written directly)
//   stat, short_list : null, or 2 words (zeroed by the caller) + [n_frames] words: see the end of the kernel
// The chain of groups is sequential, but between two groups that carry the flag nothing changes: run length and
// smallidx stay, so every group is the same number of bits, S = width + 1 + (run / 3) smallidx, and the same number of
// atoms.  The 64 lanes therefore look at the flag bits of the NEXT 64 groups at once, assuming no flag in between —
// lane k at bit pos + width + k S —, the first lane that sees a flag (ballot) tells how far the assumption held, the
// wave moves over those k groups in one step and through the flagged group behind them by hand.  A stream without
// runs (one flag in a hundred groups) goes by 64 groups a step, water (every group alike) too; a stream that changes
// its run length at every other group still takes two groups a step, and a frame is a wave, not a lane: a batch of a
// few thousand frames fills the machine.
__global__ __launch_bounds__(256) void k_xtc_scan(const uint8_t *__restrict__ blob, unsigned long long blob_bytes,
                                                  const gorder_xtc_frame_t *__restrict__ frames, uint32_t n_frames,
                                                  uint32_t natoms, const int32_t *__restrict__ slot_of, uint32_t n_stop,
                                                  float *__restrict__ out, uint32_t n_out, uint32_t *err, uint32_t *stat,
                                                  uint32_t *short_list, XtcCheckpoint *__restrict__ cp, uint32_t n_chunks) {
    // the stretch of the frame's stream the wave is looking at, in LDS (a step reads two bytes per lane within
    // 64 groups of the current one: 320 bytes for groups of 40 bits, 6.4 KB for the widest possible ones)
    constexpr uint32_t kWin = 8192;
    __shared__ __attribute__((aligned(16))) uint8_t l_win[4][kWin + 16];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t fr = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (fr >= n_frames) return;                  // (uniform per wave; no workgroup barrier in this kernel)
    uint8_t *win = l_win[threadIdx.x >> 6];
    const gorder_xtc_frame_t d = frames[fr];
    // the frame's region of the blob: the block rounded up to whole 64-byte pieces plus one piece of zeros
    const unsigned long long region = (((unsigned long long)d.n_bytes + 63ull) & ~63ull) + 64ull;
    bool bad = (d.offset & 63ull) != 0ull || d.offset + region > blob_bytes;
    const uint8_t *p = blob + (bad ? 0ull : d.offset);
    const uint32_t last_byte = bad ? 62u : (uint32_t)(region > 0xfffffffeull ? 0xfffffffeull : region) - 2u;   // two bytes are read at a time
    XtcCheckpoint *mycp = cp + (size_t)fr * (n_chunks + 1u);

    if (natoms <= 9u) {      // uncompressed small systems: big-endian floats, written here; the chunks are empty
        float *o = out + (size_t)fr * n_out * 3u;
        for (uint32_t t = lane; t < n_stop; t += 64u) {
            const int32_t slot = slot_of ? slot_of[t] : (int32_t)t;
            if (slot < 0 || (uint32_t)slot >= n_out || bad || d.n_bytes < 12u * natoms) continue;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(p) + 3u * t;
            for (int c = 0; c < 3; c++) o[3u * (size_t)slot + c] = __uint_as_float(__builtin_bswap32(w[c]));
        }
        for (uint32_t c = lane; c <= n_chunks; c += 64u) mycp[c] = XtcCheckpoint{0ull, n_stop, (uint32_t)kXtcFirstIdx};
        if (bad && lane == 0u) raise_error(err, GORDER_ERR_TRAJECTORY_FORMAT, fr, kStageBox);
        return;
    }
    int smallidx = d.smallidx;
    if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) { bad = true; smallidx = kXtcFirstIdx; }
    // bits of a full-width atom: one mixed-radix number, or three fields
    const uint32_t width = d.bitsize ? d.bitsize : (d.bitsizeint & 0xffu) + ((d.bitsizeint >> 8) & 0xffu) + ((d.bitsizeint >> 16) & 0xffu);
    // (bit positions in 32 bits: a frame of 256 MB or more is refused here — gorder_xtc_pack_window's blob could not
    // hold many of them either)
    if (d.n_bytes >= (1u << 28)) bad = true;
    // widths no XTC writer produces (the table is the caller's: gorder_hip_xtc_decode is public): a mixed-radix atom of
    // more than 72 bits, a separate field of more than 32, or no bits at all — 63 groups of such a width would not
    // fit the window below, and the chunk decoder's bit reader takes at most 64 bits at a time
    if (d.bitsize > 72u || width == 0u ||
        (!d.bitsize && ((d.bitsizeint & 0xffu) > 32u || ((d.bitsizeint >> 8) & 0xffu) > 32u || ((d.bitsizeint >> 16) & 0xffu) > 32u)))
        bad = true;
    // the walk's state is the same in every lane
    uint32_t pos = 0;                // bits of the stream before the current group
    uint32_t atoms = 0, next_cp = 0;
    uint32_t run3 = 0;               // small atoms behind a full-width atom (run / 3); a group without the flag keeps the last value
    // checkpoints at the starts of the groups j < k of a stretch of equal groups (S bits, A atoms each) from (pos, atoms):
    // checkpoint c goes to the first group with at least c kXtcChunk atoms before it; lane i looks after checkpoint next_cp + i
    auto checkpoints = [&](uint32_t k, uint32_t S, uint32_t A) {
        const uint32_t c = next_cp + lane, need = c * kXtcChunk;
        const uint32_t j = need <= atoms ? 0u : (need - atoms + A - 1u) / A;
        const bool mine = c <= n_chunks && j < k;
        if (mine) mycp[c] = XtcCheckpoint{(unsigned long long)pos + (unsigned long long)j * S, atoms + j * A, (uint32_t)smallidx | ((3u * run3) << 8)};
        next_cp += (uint32_t)__popcll(__ballot(mine));
    };
    uint32_t wbase = 0xffffffffu;    // first byte of the stream in the window (a multiple of 64); nothing loaded yet
    const unsigned long long region_last16 = bad ? 0ull : region - 16ull;       // the region's last 16 bytes are zeros
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    while (!bad && atoms < n_stop) {
        const uint32_t A = 1u + run3, S = width + 1u + run3 * (uint32_t)smallidx;
        {   // the window must hold the bytes every lane reads in this step
            const uint32_t first = pos >> 3;
            const unsigned long long need_end = (((unsigned long long)pos + width + 63ull * S) >> 3) + 2ull;
            if (wbase == 0xffffffffu || first < wbase || need_end > (unsigned long long)wbase + kWin) {      // (uniform)
                wbase = first & ~63u;
                u32x4 v[kWin / 1024u];
#pragma unroll
                for (uint32_t t = 0; t < kWin / 1024u; t++) {
                    const unsigned long long off = (unsigned long long)wbase + (t * 64u + lane) * 16u;
                    v[t] = *reinterpret_cast<const u32x4 *>(p + (off < region_last16 ? off : region_last16));
                }
#pragma unroll
                for (uint32_t t = 0; t < kWin / 1024u; t++) *reinterpret_cast<u32x4 *>(win + (t * 64u + lane) * 16u) = v[t];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        // groups still wanted: the walk ends with the first group boundary at or behind n_stop atoms (an integer division
        // costs as much as the rest of the step: only near the end)
        const uint32_t left = n_stop - atoms;
        const uint32_t k_want = left >= 64u * A ? 64u : (left + A - 1u) / A;
        // lane k: the flag bit of group k and the five bits behind it, IF the groups before it are all like this one
        const unsigned long long fp = (unsigned long long)pos + width + (unsigned long long)lane * S;
        const unsigned long long by64 = fp >> 3;
        const uint32_t by = (uint32_t)by64 - wbase;                                   // (inside the window by the check above)
        const uint32_t two = ((uint32_t)win[by] << 8) | (uint32_t)win[by + 1u];
        const uint32_t six = by64 < last_byte ? (two >> (10u - ((uint32_t)fp & 7u))) & 63u : 0u;     // (past the block: zeros)
        const unsigned long long flagged = __ballot((six & 32u) != 0u && lane < k_want);
        const uint32_t k_flag = flagged ? (uint32_t)__builtin_ctzll(flagged) : 64u;
        const uint32_t k = min(k_flag, k_want);               // groups without a flag from here on
        // (a checkpoint falls on one of them only if the last of them starts with enough atoms before it)
        if (k > 0u && next_cp <= n_chunks && next_cp * kXtcChunk <= atoms + (k - 1u) * A) checkpoints(k, S, A);
        pos += k * S;
        atoms += k * A;
        if (k_flag < 64u && k == k_flag) {      // the group with the flag: its run length and the step of smallidx
            const uint32_t s6 = (uint32_t)__builtin_amdgcn_readlane((int)six, (int)k_flag);
            const uint32_t r5 = s6 & 31u, q3 = (r5 * 43u) >> 7;       // r5 / 3 for r5 < 64
            const int is_smaller = (int)(r5 - 3u * q3) - 1;
            run3 = q3;
            const uint32_t A1 = 1u + run3, S1 = width + 6u + run3 * (uint32_t)smallidx;
            if (next_cp <= n_chunks && next_cp * kXtcChunk <= atoms) checkpoints(1u, S1, A1);
            pos += S1;
            atoms += A1;
            smallidx += is_smaller;
            if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) { bad = true; smallidx = kXtcFirstIdx; }
        }
        if (atoms > natoms || pos > 0x7ffffff0u) bad = true;
    }
    // Read into the padding?  A frame of which only a leading part was copied (bit 1 of kind) is then SHORT, not corrupt
    // (whatever else went wrong while it walked zeros): it goes on the caller's list and is decoded elsewhere.
    // stat[0] = short frames, stat[1] = max over frames of bytes needed / bytes given, in units of 2^-16.
    const bool over = (unsigned long long)pos > 8ull * d.n_bytes;
    // a frame that cannot be decoded gets empty chunks; the checkpoints not reached are where the walk ended
    if (bad || over) { next_cp = 0; atoms = 0; pos = 0; }
    for (uint32_t c = next_cp + lane; c <= n_chunks; c += 64u) mycp[c] = XtcCheckpoint{(unsigned long long)pos, atoms, (uint32_t)smallidx | ((3u * run3) << 8)};
    if (lane == 0u) {
        if (stat) {
            const unsigned long long q16 = ((((unsigned long long)pos + 7ull) >> 3) << 16) / (d.n_bytes ? d.n_bytes : 1u);
            atomicMax(&stat[1], (uint32_t)(q16 > 0xffffffffull ? 0xffffffffull : q16));
        }
        if (over && (d.kind & 2u) && stat && short_list) short_list[atomicAdd(&stat[0], 1u)] = fr;
        else if (bad || over) raise_error(err, GORDER_ERR_TRAJECTORY_FORMAT, fr, kStageBox);
    }
}

// ---- pass 2: the chunks -----------------------------------------------------------------------------------------------
// grid = ceil(n_frames * n_chunks / 64) blocks of ONE wave; lane = (frame, chunk), consecutive lanes = consecutive chunks.
//   slot_of : [natoms] output slot of an atom or -1 (null: every atom, slot = atom)
//   out     : [n_frames][n_out][3]
__global__ __launch_bounds__(64) void k_xtc_chunks(const uint8_t *__restrict__ blob, unsigned long long blob_bytes,
                                                   const gorder_xtc_frame_t *__restrict__ frames, uint32_t n_frames,
                                                   uint32_t natoms, const int32_t *__restrict__ slot_of, uint32_t n_stop,
                                                   float *__restrict__ out, uint32_t n_out,
                                                   const XtcCheckpoint *__restrict__ cp, uint32_t n_chunks) {
    __shared__ uint32_t l_magic[kXtcLastIdx];
    __shared__ unsigned long long l_recip[kXtcLastIdx];
    __shared__ double l_inv[kXtcLastIdx];
    const uint32_t lane = threadIdx.x;
    for (uint32_t k = lane; k < (uint32_t)kXtcLastIdx; k += 64u) {
        const uint32_t m = kXtcMagic[k];
        l_magic[k] = m;
        // floor(2^64 / m): a power of two divides 2^64 exactly, for anything else it is floor((2^64 - 1) / m)
        l_recip[k] = m == 0u ? 0ull : ((m & (m - 1u)) == 0u ? 1ull << (64 - __builtin_ctz(m)) : ~0ull / m);
        l_inv[k] = m == 0u ? 0.0 : 1.0 / (double)m;
    }
    __syncthreads();
    const unsigned long long total = (unsigned long long)n_frames * n_chunks;
    const unsigned long long item = (unsigned long long)blockIdx.x * 64u + lane;
    const bool live = item < total;
    const unsigned long long it = live ? item : total - 1ull;
    const uint32_t fr = (uint32_t)(it / n_chunks), ch = (uint32_t)(it - (unsigned long long)fr * n_chunks);
    const gorder_xtc_frame_t d = frames[fr];
    const XtcCheckpoint ca = cp[(size_t)fr * (n_chunks + 1u) + ch], cb = cp[(size_t)fr * (n_chunks + 1u) + ch + 1u];
    const float inv_p = d.inv_precision;
    const unsigned long long region = (((unsigned long long)d.n_bytes + 63ull) & ~63ull) + 64ull;
    const bool bad_frame = (d.offset & 63ull) != 0ull || d.offset + region > blob_bytes || natoms <= 9u;
    const uint8_t *p = blob + (bad_frame ? 0ull : d.offset);
    const unsigned long long last_piece = bad_frame ? 0ull : region - 64ull;
    const uint32_t atom0 = ca.atom;
    // atoms of this chunk (a checkpoint lies within a group of the chunk's nominal start: never more than kXtcChunk + 8)
    uint32_t n_at = (live && !bad_frame && cb.atom >= ca.atom && cb.atom - ca.atom <= kXtcChunk + 16u) ? cb.atom - ca.atom : 0u;
    uint32_t n_max = n_at;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, off, 64));
    if (n_max == 0u) return;                     // (uniform; no workgroup barrier below this line)

    // Decoded atoms go through LDS: a lane writing its chunk's atom straight to memory is one 12-byte piece in each of
    // 64 places per store instruction.  Instead a lane parks atom i of its chunk at l_atoms[i mod 16] and, once 8
    // consecutive atoms are final, the wave writes them chunk by chunk: 24 lanes = 8 atoms x 3 coordinates = one
    // contiguous 96-byte piece of a frame when the analysed atoms are consecutive in the file (any other order is still
    // correct, only less coalesced), two chunks per store instruction.  The slots of those atoms (each lane looks up its
    // own, one service point ahead) and the chunks' places in `out` go through LDS as well.
    __shared__ float l_atoms[2u * kXtcGroup * 3u * kXtcPitch];
    __shared__ int32_t l_slot[kXtcGroup * kXtcPitch];
    __shared__ unsigned long long l_obase[64];
    l_obase[lane] = (unsigned long long)fr * n_out * 3u;
    auto put = [&](const int (&c)[3], uint32_t idx) {        // idx (atom of the chunk) is wave-uniform
        float *q = l_atoms + (idx & (2u * kXtcGroup - 1u)) * 3u * kXtcPitch + lane;
        q[0] = (float)c[0] * inv_p;
        q[kXtcPitch] = (float)c[1] * inv_p;
        q[2u * kXtcPitch] = (float)c[2] * inv_p;
    };
    auto slot_of_local = [&](uint32_t t) -> int32_t {        // output slot of atom t of MY chunk (-1: none)
        const uint32_t idx = atom0 + t;
        const bool want = t < n_at && idx < n_stop;
        if (!slot_of) return want ? (int32_t)idx : -1;
        const int32_t sl = slot_of[want ? idx : 0u];
        return want ? sl : -1;
    };
    int32_t snext[kXtcGroup];
#pragma unroll
    for (uint32_t j = 0; j < kXtcGroup; j++) snext[j] = slot_of_local(j);
    const uint32_t fl_half = lane >= 3u * kXtcGroup ? 1u : 0u, fl_l = lane - fl_half * 3u * kXtcGroup;
    const uint32_t fl_j = fl_l / 3u, fl_c = fl_l - 3u * fl_j;
    auto flush = [&](uint32_t first) {    // atoms first .. first + 7 of every chunk, all final; their slots are in snext
#pragma unroll
        for (uint32_t j = 0; j < kXtcGroup; j++) l_slot[j * kXtcPitch + lane] = snext[j];
        // the wave reads what its own lanes wrote: LDS instructions of one wave execute in order, the fences only keep
        // the compiler from moving the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lanes 0..23: (atom, coordinate) of the even chunks, lanes 24..47: of the odd chunks — 96 contiguous bytes each
        if (lane < 6u * kXtcGroup) {
            const float *q = l_atoms + (((first + fl_j) & (2u * kXtcGroup - 1u)) * 3u + fl_c) * kXtcPitch;
            const int32_t *sl = l_slot + fl_j * kXtcPitch;
#pragma unroll 4
            for (uint32_t f = fl_half; f < 64u; f += 2u) {
                const int32_t slot = sl[f];
                if (slot >= 0 && (uint32_t)slot < n_out) out[l_obase[f] + (size_t)(uint32_t)slot * 3u + fl_c] = q[f];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- the input side: each lane's stream goes through a 256-byte ring in LDS, refilled in 64-byte pieces at the
    // same points where the output is flushed.  A piece is REQUESTED at one service point (global loads into registers)
    // and LANDED in the ring at the next one, eight atoms later: no instruction of the loop ever waits for memory.  A
    // lane uses at most 78 bytes between two service points (8 atoms of 72 + 6 bits) and gets up to 128, a corrupt
    // stream just runs into the region's zero piece (the piece ADDRESS is clamped).  The lane's stream starts at the
    // 64-byte piece that holds its checkpoint's bit.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned long long l_in[64u * kXtcRingPitch];
    unsigned long long *ring = l_in + lane * kXtcRingPitch;
    const unsigned long long byte0 = (ca.bit >> 9) * 64ull;
    uint32_t wr = 0;                 // bytes of the lane's stream (from byte0 on) landed in the ring (a multiple of 64)
    uint32_t n_req = 0;              // pieces requested at the last service point (0..2)
    u32x4 pc[8];
    auto piece_at = [&](uint32_t off) {
        const unsigned long long at = byte0 + off;
        return reinterpret_cast<const u32x4 *>(p + (at < last_piece ? at : last_piece));
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
    b.open(ring, (uint32_t)(ca.bit & 511ull));
    // A service point, every kXtcGroup atoms: land what was requested at the last one, flush the finished atoms, request
    // the next pieces and the output slots of the next flush.  Every global load is consumed one service point after it
    // was issued, so the only waits on memory are for operations eight atoms old.
    auto service = [&](uint32_t first) {
        if (n_req >= 1u) land(pc);
        if (n_req == 2u) land(pc + 4);
        flush(first);
#pragma unroll
        for (uint32_t j = 0; j < kXtcGroup; j++) snext[j] = slot_of_local(first + kXtcGroup + j);
        const uint32_t room = 8u * kXtcRingWords - (wr - 8u * b.rd);        // bytes of the ring not holding unread stream
        n_req = min(2u, room / 64u);
        if (n_req >= 1u) { const u32x4 *src = piece_at(wr); pc[0] = src[0]; pc[1] = src[1]; pc[2] = src[2]; pc[3] = src[3]; }
        if (n_req == 2u) { const u32x4 *src = piece_at(wr + 64u); pc[4] = src[0]; pc[5] = src[1]; pc[6] = src[2]; pc[7] = src[3]; }
    };
    int smallidx = (int)(ca.state & 0xffu);
    if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) { n_at = 0; smallidx = kXtcFirstIdx; }
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
        if (smallidx < kXtcFirstIdx || smallidx >= kXtcLastIdx) smallidx = kXtcFirstIdx;     // (k_xtc_scan has raised the error)
        sizesmall = l_magic[smallidx];
        rsmall = l_recip[smallidx];
        sinv = l_inv[smallidx];
        smallnum = (int)(sizesmall / 2u);
    };
    int run = (int)((ca.state >> 8) & 0xffu), pend = 0;      // (a group without the flag repeats the run length before it)
    uint32_t run_left = 0;          // small atoms of the current run still to come
    bool first = false;             // the next small atom is the first of its run (it swaps places with its predecessor)
    int prev[3] = {0, 0, 0};
    uint32_t flushed = 0;           // atoms of the chunk written out so far (a multiple of kXtcGroup)
    // ---- the usual case, for the whole wave: every frame's full-width atom is ONE number of at most 53 bits (box edges
    // below ~100 000 grid steps).  A full-width atom and a small-offset atom are then the SAME computation with other
    // parameters — a mixed-radix number of nb bits with radices (s1, s2), cut out of the next 64 bits of the stream —
    // so the lanes of a wave, which are in the two states at the same time, run ONE copy of it instead of both.  The two
    // divisions are done in f64: v < 2^53 is exact as a double, floor(v * (1 / s)) is within 1 of the quotient, the
    // remainder by fma is exact and repairs it — 8 f64 operations per division where floor(2^64 / s) arithmetic needs 7
    // quarter-rate integer multiplications.
    const bool simple = d.bitsize >= 1u && d.bitsize <= 53u;
    if (__ballot(!simple) == 0ull) {
        const double ls1 = (double)d.sizeint[1], ls2 = (double)d.sizeint[2];
        const double li1 = 1.0 / ls1, li2 = 1.0 / ls2;
        for (uint32_t t = 0; t < n_max; t++) {
            if (t == flushed + kXtcGroup + 1u) {     // atoms < t - 1 are final
                service(flushed);
                flushed += kXtcGroup;
            }
            if (t >= n_at) continue;
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
                    if (t + 1u + run_left > n_at) run_left = 0;      // (a corrupt stream; k_xtc_scan has raised the error)
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
    for (uint32_t t = 0; t < n_max; t++) {
        if (t == flushed + kXtcGroup + 1u) {     // atoms < t - 1 are final
            service(flushed);
            flushed += kXtcGroup;
        }
        if (t >= n_at) continue;
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
                if (t + 1u + run_left > n_at) run_left = 0;
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
    for (; flushed < n_max; flushed += kXtcGroup) {
        flush(flushed);
#pragma unroll
        for (uint32_t j = 0; j < kXtcGroup; j++) snext[j] = slot_of_local(flushed + kXtcGroup + j);
    }
}

}  // namespace
