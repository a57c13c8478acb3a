// xtc_reader.cpp — host-side GROMACS XTC reader (include/gorder_xtc.h).  Pure C++17, no HIP.
//
// Wire format (published GROMACS "xtc" / xdr3dcoord): big-endian XDR.  Per frame:
//   int32 magic = 1995, int32 natoms, int32 step, float32 time, float32 box[3][3],
//   int32 natoms; if natoms <= 9: natoms*3 raw float32.  Otherwise:
//   float32 precision, int32 minint[3], int32 maxint[3], int32 smallidx, int32 nbytes,
//   nbytes of bit stream (padded to a multiple of 4).
// The bit stream packs each atom's three integers (coordinate * precision, offset by minint) either
// in full width or, for runs of neighbouring atoms, as small differences in a mixed-radix number
// whose radix comes from the `magicints` table.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <string>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../include/gorder_xtc.h"

namespace {

const int kMagicInts[] = {
    0,       0,       0,       0,       0,       0,       0,       0,       0,       8,        10,       12,
    16,      20,      25,      32,      40,      50,      64,      80,      101,     128,      161,      203,
    256,     322,     406,     512,     645,     812,     1024,    1290,    1625,    2048,     2580,     3250,
    4096,    5060,    6501,    8192,    10321,   13003,   16384,   20642,   26007,   32768,    41285,    52015,
    65536,   82570,   104031,  131072,  165140,  208063,  262144,  330280,  416127,  524287,   660561,   832255,
    1048576, 1321122, 1664510, 2097152, 2642245, 3329021, 4194304, 5284491, 6658042, 8388607,  10568983, 13316085,
    16777216};
constexpr int kFirstIdx = 9;
constexpr int kLastIdx = (int)(sizeof(kMagicInts) / sizeof(kMagicInts[0]));

// The compressed block is one MSB-first bit stream.  The reader keeps an absolute bit position and takes up to 57
// bits at a time from one unaligned big-endian 64-bit load (the caller pads the block with 8 zero bytes).
struct BitReader {
    const uint8_t *p;
    size_t n, bitpos = 0;
    bool overrun = false;

    static uint64_t load_be64(const uint8_t *q) {
        uint64_t w;
        memcpy(&w, q, 8);
        return __builtin_bswap64(w);
    }
    uint64_t bits57(int nbits) {      // 0 <= nbits <= 57
        if (nbits == 0) return 0;
        const size_t byte = bitpos >> 3;
        if (byte >= n) { overrun = true; return 0; }    // n bytes of payload + 8 bytes of padding are readable
        const uint64_t w = load_be64(p + byte) << (bitpos & 7u);
        bitpos += (size_t)nbits;
        return w >> (64 - nbits);
    }
    uint32_t bits(int nbits) { return (uint32_t)bits57(nbits); }          // nbits <= 32
    // three integers packed as one mixed-radix number of `nbits` bits with radices sizes[0..2].  The number is
    // stored in chunks of 8 bits, FIRST chunk least significant (the last, partial chunk is the top): for up to 64
    // bits that is a byte swap of the stream bits, then two 64-bit divisions; wider numbers (boxes beyond ~2 million
    // grid steps per edge) take the byte-wise long division of the original algorithm.
    // recip[k] = floor(2^64 / sizes[k]) (reciprocal_of), k = 1, 2
    void ints(int nbits, const uint32_t sizes[3], const uint64_t recip[3], int out[3]) {
        if (nbits <= 64) {
            const int m = (nbits - 1) / 8, rem = nbits - 8 * m;      // m full chunks, then `rem` (1..8) bits
            uint64_t v = 0;
            if (m > 0) {
                // the m leading bytes of the stream, first byte lowest
                const int lo_bits = 8 * m;
                uint64_t w = lo_bits <= 57 ? bits57(lo_bits) : ((bits57(32) << (lo_bits - 32)) | bits57(lo_bits - 32));
                v = __builtin_bswap64(w << (64 - lo_bits));
            }
            v |= bits57(rem) << (8 * m);
            // two divisions by the radices — as multiplications by their precomputed reciprocals (a 64-bit hardware
            // division costs several times the rest of the atom): q = floor(v * floor(2^64 / s) / 2^64) is at most
            // two short of floor(v / s), the remainder test repairs it
            const uint64_t s2 = sizes[2], s1 = sizes[1];
            uint64_t q2 = (uint64_t)(((unsigned __int128)v * recip[2]) >> 64);
            uint64_t r2 = v - q2 * s2;
            if (r2 >= s2) { r2 -= s2; q2++; }
            if (r2 >= s2) { r2 -= s2; q2++; }
            out[2] = (int)r2;
            uint64_t q1 = (uint64_t)(((unsigned __int128)q2 * recip[1]) >> 64);
            uint64_t r1 = q2 - q1 * s1;
            if (r1 >= s1) { r1 -= s1; q1++; }
            if (r1 >= s1) { r1 -= s1; q1++; }
            out[1] = (int)r1;
            out[0] = (int)(uint32_t)q1;
            return;
        }
        uint32_t bytes[32];
        bytes[1] = bytes[2] = bytes[3] = 0;
        int nbytes = 0;
        while (nbits > 8) {
            bytes[nbytes++] = bits(8);
            nbits -= 8;
        }
        if (nbits > 0) bytes[nbytes++] = bits(nbits);
        for (int i = 2; i > 0; i--) {
            uint32_t num = 0;
            for (int j = nbytes - 1; j >= 0; j--) {
                num = (num << 8) | bytes[j];
                const uint32_t q = num / sizes[i];
                bytes[j] = q;
                num -= q * sizes[i];
            }
            out[i] = (int)num;
        }
        out[0] = (int)(bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | (bytes[3] << 24));
    }
};

inline uint64_t reciprocal_of(uint32_t s) {      // floor(2^64 / s); s = 1 saturates (then q = v - 1 short by one: repaired)
    return s <= 1u ? ~0ull : (uint64_t)((((unsigned __int128)1) << 64) / s);
}
// the reciprocals of the "magic" radices of the small-offset runs, made once
struct MagicRecips {
    uint64_t r[kLastIdx];
    MagicRecips() { for (int i = 0; i < kLastIdx; i++) r[i] = kMagicInts[i] > 0 ? reciprocal_of((uint32_t)kMagicInts[i]) : 0; }
};
const MagicRecips kMagicRecips;

int size_of_int(uint32_t size) {
    uint32_t num = 1;
    int nbits = 0;
    while (size >= num && nbits < 32) {
        nbits++;
        num <<= 1;
    }
    return nbits;
}

int size_of_ints(const uint32_t sizes[3]) {
    uint32_t bytes[32];
    int nbytes = 1;
    bytes[0] = 1;
    for (int i = 0; i < 3; i++) {
        uint32_t tmp = 0;
        int b = 0;
        for (; b < nbytes; b++) {
            tmp = bytes[b] * sizes[i] + tmp;
            bytes[b] = tmp & 0xff;
            tmp >>= 8;
        }
        while (tmp != 0) {
            bytes[b++] = tmp & 0xff;
            tmp >>= 8;
        }
        nbytes = b;
    }
    uint32_t num = 1;
    nbytes--;
    int nbits = 0;
    while (bytes[nbytes] >= num) {
        nbits++;
        num *= 2;
    }
    return nbits + nbytes * 8;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
float bef(const uint8_t *p) {
    const uint32_t u = be32(p);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

double bed(const uint8_t *p) {
    uint64_t u = 0;
    for (int k = 0; k < 8; k++) u = (u << 8) | p[k];
    double d;
    memcpy(&d, &u, 8);
    return d;
}

}  // namespace

// A read-only mapping of a trajectory file: the block copies of gorder_xtc_pack_window* read the page cache through it
// (streaming stores into the staging blob) instead of one pread per block — the kernel's copy_to_user writes the
// destination through the cache, and with sixteen threads those read-for-ownership misses, not the copy, are the
// ceiling (tools/microbench/copy_bench.cpp: 5 GB/s with pread or memcpy on 8 threads, 32-38 GB/s with non-temporal
// stores).  Shared with the copy jobs of a pool so that the reader may be closed before they are done.
struct FileMap {
    const uint8_t *base = nullptr;
    size_t size = 0;
    ~FileMap() { if (base) munmap(const_cast<uint8_t *>(base), size); }
};

struct gorder_xtc_reader {
    std::string path;                 // for the worker threads of gorder_xtc_read_window_mt (own file handles)
    FILE *fp = nullptr;
    bool trr = false;                 // GROMACS TRR (magic 1993, uncompressed reals) instead of XTC (magic 1995)
    bool gro = false;                 // multi-frame GRO text (groan_rs GroReader, common.rs:322-333)
    uint32_t natoms = 0;
    std::vector<uint32_t> group;      // atoms to convert (empty = all)
    std::vector<int32_t> slot_of;     // atom -> output slot or -1 (only when group given)
    uint32_t n_needed = 0;            // atoms to decompress per frame: up to the last atom of the group (the bit stream is
                                      // sequential, but nothing after that atom is wanted — e.g. the water behind the lipids)
    std::vector<uint8_t> buf;
    std::vector<int> ints;            // decoded integer coordinates of one frame
    std::shared_ptr<struct FileMap> map;   // gorder_xtc_pack_window*: the file mapped once (made at the first window)
    bool map_tried = false;
};

namespace {

bool read_exact(FILE *fp, void *dst, size_t n) { return fread(dst, 1, n, fp) == n; }

// Decode the compressed block into integer coordinates for all atoms.
int decode_ints(gorder_xtc_reader *r, const int minint[3], const int maxint[3], int smallidx, const uint8_t *data,
                size_t nbytes) {
    const uint32_t natoms = r->natoms;
    r->ints.resize((size_t)natoms * 3);
    uint32_t sizeint[3];
    int bitsizeint[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        const int64_t s = (int64_t)maxint[k] - (int64_t)minint[k] + 1;
        if (s <= 0 || s > 0xffffffffll) return GORDER_XTC_ERR_FORMAT;
        sizeint[k] = (uint32_t)s;
    }
    int bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int k = 0; k < 3; k++) bitsizeint[k] = size_of_int(sizeint[k]);
        bitsize = 0;
    } else {
        bitsize = size_of_ints(sizeint);
    }
    if (smallidx < kFirstIdx || smallidx >= kLastIdx) return GORDER_XTC_ERR_FORMAT;
    int smaller = kMagicInts[smallidx > kFirstIdx ? smallidx - 1 : kFirstIdx] / 2;
    int smallnum = kMagicInts[smallidx] / 2;
    uint32_t sizesmall[3] = {(uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx]};
    const uint64_t recip_big[3] = {reciprocal_of(sizeint[0]), reciprocal_of(sizeint[1]), reciprocal_of(sizeint[2])};
    uint64_t recip_small[3] = {kMagicRecips.r[smallidx], kMagicRecips.r[smallidx], kMagicRecips.r[smallidx]};

    BitReader br{data, nbytes};
    int *out = r->ints.data();
    uint32_t i = 0;
    int run = 0;
    const uint32_t n_stop = r->n_needed ? std::min(r->n_needed, natoms) : natoms;
    while (i < n_stop) {
        int cur[3];
        if (bitsize == 0) {
            for (int k = 0; k < 3; k++) cur[k] = (int)br.bits(bitsizeint[k]);
        } else {
            br.ints(bitsize, sizeint, recip_big, cur);
        }
        i++;
        for (int k = 0; k < 3; k++) cur[k] += minint[k];
        int prev[3] = {cur[0], cur[1], cur[2]};
        const uint32_t flag = br.bits(1);
        int is_smaller = 0;
        if (flag == 1) {
            run = (int)br.bits(5);
            is_smaller = run % 3;
            run -= is_smaller;
            is_smaller--;
        }
        if (run > 0) {
            if (i + (uint32_t)(run / 3) > natoms) return GORDER_XTC_ERR_FORMAT;
            for (int k = 0; k < run; k += 3) {
                int d[3];
                br.ints(smallidx, sizesmall, recip_small, d);
                i++;
                for (int c = 0; c < 3; c++) cur[c] = d[c] + prev[c] - smallnum;
                if (k == 0) {
                    // the first atom of a run is stored AFTER the second one (water: O after H)
                    for (int c = 0; c < 3; c++) { const int t = cur[c]; cur[c] = prev[c]; prev[c] = t; }
                    out[0] = prev[0]; out[1] = prev[1]; out[2] = prev[2];
                    out += 3;
                } else {
                    prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
                }
                out[0] = cur[0]; out[1] = cur[1]; out[2] = cur[2];
                out += 3;
            }
        } else {
            out[0] = cur[0]; out[1] = cur[1]; out[2] = cur[2];
            out += 3;
        }
        smallidx += is_smaller;
        if (smallidx < kFirstIdx || smallidx >= kLastIdx) return GORDER_XTC_ERR_FORMAT;
        if (is_smaller < 0) {
            smallnum = smaller;
            smaller = smallidx > kFirstIdx ? kMagicInts[smallidx - 1] / 2 : 0;
        } else if (is_smaller > 0) {
            smaller = smallnum;
            smallnum = kMagicInts[smallidx] / 2;
        }
        sizesmall[0] = sizesmall[1] = sizesmall[2] = (uint32_t)kMagicInts[smallidx];
        recip_small[0] = recip_small[1] = recip_small[2] = kMagicRecips.r[smallidx];
        if (br.overrun) return GORDER_XTC_ERR_FORMAT;
    }
    if (br.bitpos > 8 * nbytes) return GORDER_XTC_ERR_FORMAT;     // read into the padding: truncated block
    return GORDER_XTC_OK;                                         // (a run may have carried `out` past n_stop: the buffer holds all atoms)
}

// ---- TRR ------------------------------------------------------------------------------------------
// GROMACS full-precision trajectory (the reference reads it through groan_rs' TrrReader, common.rs:306-320).
// XDR, big-endian.  Frame = header {magic 1993, version string "GMX_trn_file", 13 ints: ir, e, box, vir, pres,
// top, sym, x, v, f sizes in bytes, natoms, step, nre; then t and lambda as reals} + box + (virial, pressure)
// + positions (+ velocities, forces).  A real is a float when box_size / 9 (or x_size / 3N) is 4, a double when 8.
struct TrrHeader {
    uint32_t sizes[10];   // ir e box vir pres top sym x v f
    uint32_t natoms;
    int32_t step;
    uint32_t real_size;
    double t;
};

int trr_read_header(gorder_xtc_reader *r, TrrHeader &h) {
    uint8_t b[8];
    const size_t got = fread(b, 1, 8, r->fp);
    if (got == 0) return GORDER_XTC_EOF;
    if (got != 8 || be32(b) != 1993u) return GORDER_XTC_ERR_FORMAT;
    const uint32_t slen = be32(b + 4);                       // length of the version string incl. terminator
    uint8_t s4[4];
    if (slen > 128 || !read_exact(r->fp, s4, 4)) return GORDER_XTC_ERR_FORMAT;
    const uint32_t n = be32(s4), padded = (n + 3u) & ~3u;
    uint8_t str[132];
    if (n > 128 || !read_exact(r->fp, str, padded)) return GORDER_XTC_ERR_FORMAT;
    uint8_t ints[13 * 4];
    if (!read_exact(r->fp, ints, sizeof(ints))) return GORDER_XTC_ERR_FORMAT;
    for (int k = 0; k < 10; k++) h.sizes[k] = be32(ints + 4 * k);
    h.natoms = be32(ints + 40);
    h.step = (int32_t)be32(ints + 44);
    if (h.sizes[2]) h.real_size = h.sizes[2] / 9u;
    else if (h.sizes[7] && h.natoms) h.real_size = h.sizes[7] / (3u * h.natoms);
    else if (h.sizes[8] && h.natoms) h.real_size = h.sizes[8] / (3u * h.natoms);
    else if (h.sizes[9] && h.natoms) h.real_size = h.sizes[9] / (3u * h.natoms);
    else h.real_size = 4;
    if (h.real_size != 4 && h.real_size != 8) return GORDER_XTC_ERR_FORMAT;
    uint8_t tl[16];
    if (!read_exact(r->fp, tl, 2 * h.real_size)) return GORDER_XTC_ERR_FORMAT;
    h.t = h.real_size == 4 ? (double)bef(tl) : bed(tl);
    return GORDER_XTC_OK;
}

int trr_next(gorder_xtc_reader *r, float *xyz, float *box9, int64_t *step, float *time_ps, float *precision) {
    for (;;) {
        TrrHeader h{};
        const int st = trr_read_header(r, h);
        if (st != GORDER_XTC_OK) return st;
        if (h.natoms != r->natoms) return GORDER_XTC_ERR_FORMAT;
        const size_t body = (size_t)h.sizes[0] + h.sizes[1] + h.sizes[2] + h.sizes[3] + h.sizes[4] + h.sizes[5] +
                            h.sizes[6] + h.sizes[7] + h.sizes[8] + h.sizes[9];
        if (h.sizes[7] == 0) {           // a frame without positions (velocities / forces only): not a frame for us
            if (fseek(r->fp, (long)body, SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
            continue;
        }
        if (h.sizes[7] != (size_t)h.natoms * 3 * h.real_size) return GORDER_XTC_ERR_FORMAT;
        if (step) *step = h.step;
        if (time_ps) *time_ps = (float)h.t;
        if (precision) *precision = 0.0f;
        if (fseek(r->fp, (long)(h.sizes[0] + h.sizes[1]), SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
        if (h.sizes[2]) {
            uint8_t b[72];
            if (h.sizes[2] > sizeof(b) || !read_exact(r->fp, b, h.sizes[2])) return GORDER_XTC_ERR_FORMAT;
            if (box9) for (int k = 0; k < 9; k++) box9[k] = h.real_size == 4 ? bef(b + 4 * k) : (float)bed(b + 8 * k);
        } else if (box9) {
            for (int k = 0; k < 9; k++) box9[k] = 0.0f;
        }
        if (fseek(r->fp, (long)(h.sizes[3] + h.sizes[4] + h.sizes[5] + h.sizes[6]), SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
        if (!xyz) {
            if (fseek(r->fp, (long)(h.sizes[7] + h.sizes[8] + h.sizes[9]), SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
            return GORDER_XTC_OK;
        }
        r->buf.resize(h.sizes[7]);
        if (!read_exact(r->fp, r->buf.data(), h.sizes[7])) return GORDER_XTC_ERR_FORMAT;
        const uint8_t *q = r->buf.data();
        const bool all = r->group.empty();
        const size_t nout = all ? h.natoms : r->group.size();
        for (size_t k = 0; k < nout; k++) {
            const size_t a = all ? k : r->group[k];
            for (int c = 0; c < 3; c++)
                xyz[3 * k + c] = h.real_size == 4 ? bef(q + 4 * (3 * a + c)) : (float)bed(q + 8 * (3 * a + c));
        }
        if (fseek(r->fp, (long)(h.sizes[8] + h.sizes[9]), SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
        return GORDER_XTC_OK;
    }
}

}  // namespace


// ---- GRO -------------------------------------------------------------------------------------------
// Text frames: title line (GROMACS writes "... t= <ps> step= <n>"), atom count, one fixed-column line per atom
// (positions from column 20 in three fields of equal width, "%8.3f" by default; wider fields for more decimals;
// velocities may follow), then the box line: v1(x) v2(y) v3(z) [v1(y) v1(z) v2(x) v2(z) v3(x) v3(y)].
namespace {
bool gro_line(FILE *fp, std::string &line) {
    line.clear();
    char buf[512];
    while (fgets(buf, sizeof(buf), fp)) {
        line += buf;
        if (!line.empty() && line.back() == '\n') { line.pop_back(); if (!line.empty() && line.back() == '\r') line.pop_back(); return true; }
    }
    return !line.empty();
}
int gro_natoms(FILE *fp, uint32_t &natoms) {       // reads the first two lines
    std::string line;
    if (!gro_line(fp, line) || !gro_line(fp, line)) return GORDER_XTC_ERR_FORMAT;
    char *end = nullptr;
    const long n = strtol(line.c_str(), &end, 10);
    if (end == line.c_str() || n <= 0 || n > 0x7fffffffl) return GORDER_XTC_ERR_FORMAT;
    natoms = (uint32_t)n;
    return GORDER_XTC_OK;
}
int gro_next(gorder_xtc_reader *r, float *xyz, float *box9, int64_t *step, float *time_ps, float *precision) {
    std::string line;
    if (!gro_line(r->fp, line)) return GORDER_XTC_EOF;
    if (line.find_first_not_of(" \t") == std::string::npos) {      // blank tail of the file
        if (!gro_line(r->fp, line)) return GORDER_XTC_EOF;
    }
    float t = 0.0f;
    long st = 0;
    const size_t pt = line.rfind("t=");
    if (pt != std::string::npos) t = strtof(line.c_str() + pt + 2, nullptr);
    const size_t ps = line.rfind("step=");
    if (ps != std::string::npos) st = strtol(line.c_str() + ps + 5, nullptr, 10);
    if (!gro_line(r->fp, line)) return GORDER_XTC_ERR_FORMAT;
    if ((uint32_t)strtol(line.c_str(), nullptr, 10) != r->natoms) return GORDER_XTC_ERR_FORMAT;
    const bool all = r->group.empty();
    size_t width = 0;
    for (uint32_t a = 0; a < r->natoms; a++) {
        if (!gro_line(r->fp, line)) return GORDER_XTC_ERR_FORMAT;
        if (!xyz) continue;
        const int32_t s = all ? (int32_t)a : r->slot_of[a];
        if (s < 0) continue;
        if (width == 0) {       // field width = distance between the decimal points of x and y
            const size_t d1 = line.find('.', 20), d2 = d1 == std::string::npos ? d1 : line.find('.', d1 + 1);
            if (d2 == std::string::npos) return GORDER_XTC_ERR_FORMAT;
            width = d2 - d1;
        }
        if (line.size() < 20 + 3 * width) return GORDER_XTC_ERR_FORMAT;
        for (int c = 0; c < 3; c++) {
            const std::string field = line.substr(20 + (size_t)c * width, width);
            char *end = nullptr;
            const float v = strtof(field.c_str(), &end);
            if (end == field.c_str()) return GORDER_XTC_ERR_FORMAT;
            xyz[3 * (size_t)s + c] = v;
        }
    }
    if (!gro_line(r->fp, line)) return GORDER_XTC_ERR_FORMAT;
    float b[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nb = 0;
    const char *p = line.c_str();
    for (; nb < 9; nb++) {
        char *end = nullptr;
        b[nb] = strtof(p, &end);
        if (end == p) break;
        p = end;
    }
    if (nb != 3 && nb != 9) return GORDER_XTC_ERR_FORMAT;
    if (box9) {
        box9[0] = b[0]; box9[4] = b[1]; box9[8] = b[2];
        box9[1] = b[3]; box9[2] = b[4]; box9[3] = b[5]; box9[5] = b[6]; box9[6] = b[7]; box9[7] = b[8];
    }
    if (step) *step = st;
    if (time_ps) *time_ps = t;
    if (precision) *precision = 0.0f;
    return GORDER_XTC_OK;
}
}  // namespace

extern "C" {

int gorder_xtc_open(const char *path, const uint32_t *group, uint32_t n_group, gorder_xtc_reader **out) {
    if (!path || !out) return GORDER_XTC_ERR_ARGUMENT;
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) return GORDER_XTC_ERR_OPEN;
    uint8_t head[8];
    if (!read_exact(fp, head, 8)) {
        fclose(fp);
        return GORDER_XTC_ERR_FORMAT;
    }
    const bool binary = be32(head) == 1995u || be32(head) == 1993u;
    uint32_t gro_atoms = 0;
    fseek(fp, 0, SEEK_SET);
    if (!binary && gro_natoms(fp, gro_atoms) != GORDER_XTC_OK) {    // neither magic: a GRO text trajectory, or nothing we read
        fclose(fp);
        return GORDER_XTC_ERR_FORMAT;
    }
    gorder_xtc_reader *r = new gorder_xtc_reader();
    r->fp = fp;
    r->path = path;
    fseek(fp, 0, SEEK_SET);
    if (!binary) {
        r->gro = true;
        r->natoms = gro_atoms;
    } else if (be32(head) == 1993u) {           // TRR: the atom count sits behind the version string
        r->trr = true;
        TrrHeader h{};
        if (trr_read_header(r, h) != GORDER_XTC_OK) {
            gorder_xtc_close(r);
            return GORDER_XTC_ERR_FORMAT;
        }
        r->natoms = h.natoms;
        fseek(fp, 0, SEEK_SET);
    } else {
        r->natoms = be32(head + 4);
    }
    if (group && n_group) {
        r->group.assign(group, group + n_group);
        r->slot_of.assign(r->natoms, -1);
        for (uint32_t k = 0; k < n_group; k++) {
            if (group[k] >= r->natoms) {
                gorder_xtc_close(r);
                return GORDER_XTC_ERR_ARGUMENT;
            }
            r->slot_of[group[k]] = (int32_t)k;
            r->n_needed = std::max(r->n_needed, group[k] + 1u);
        }
    }
    *out = r;
    return GORDER_XTC_OK;
}

void gorder_xtc_close(gorder_xtc_reader *r) {
    if (!r) return;
    if (r->fp) fclose(r->fp);
    delete r;
}

uint32_t gorder_xtc_n_atoms_file(const gorder_xtc_reader *r) { return r ? r->natoms : 0; }
uint32_t gorder_xtc_n_atoms_out(const gorder_xtc_reader *r) {
    return r ? (r->group.empty() ? r->natoms : (uint32_t)r->group.size()) : 0;
}

int gorder_xtc_next(gorder_xtc_reader *r, float *xyz, float *box9, int64_t *step, float *time_ps,
                    float *precision) {
    if (!r || !r->fp) return GORDER_XTC_ERR_ARGUMENT;
    if (r->trr) return trr_next(r, xyz, box9, step, time_ps, precision);
    if (r->gro) return gro_next(r, xyz, box9, step, time_ps, precision);
    uint8_t head[16 + 36 + 4];
    const size_t got = fread(head, 1, sizeof(head), r->fp);
    if (got == 0) return GORDER_XTC_EOF;
    if (got != sizeof(head) || be32(head) != 1995u) return GORDER_XTC_ERR_FORMAT;
    const uint32_t natoms = be32(head + 4);
    if (natoms != r->natoms || be32(head + 52) != natoms) return GORDER_XTC_ERR_FORMAT;
    if (step) *step = (int32_t)be32(head + 8);
    if (time_ps) *time_ps = bef(head + 12);
    if (box9) for (int k = 0; k < 9; k++) box9[k] = bef(head + 16 + 4 * k);
    const bool all = r->group.empty();
    if (natoms <= 9) {   // uncompressed small systems
        r->buf.resize((size_t)natoms * 12);
        if (!read_exact(r->fp, r->buf.data(), r->buf.size())) return GORDER_XTC_ERR_FORMAT;
        if (precision) *precision = 0.0f;
        if (xyz) {
            for (uint32_t a = 0; a < natoms; a++) {
                const int32_t s = all ? (int32_t)a : r->slot_of[a];
                if (s < 0) continue;
                for (int c = 0; c < 3; c++) xyz[3 * (size_t)s + c] = bef(r->buf.data() + 12 * (size_t)a + 4 * c);
            }
        }
        return GORDER_XTC_OK;
    }
    uint8_t h2[4 + 24 + 4 + 4];
    if (!read_exact(r->fp, h2, sizeof(h2))) return GORDER_XTC_ERR_FORMAT;
    const float prec = bef(h2);
    int minint[3], maxint[3];
    for (int k = 0; k < 3; k++) {
        minint[k] = (int32_t)be32(h2 + 4 + 4 * k);
        maxint[k] = (int32_t)be32(h2 + 16 + 4 * k);
    }
    const int smallidx = (int32_t)be32(h2 + 28);
    const uint32_t nbytes = be32(h2 + 32);
    const size_t padded = ((size_t)nbytes + 3) & ~(size_t)3;
    if (precision) *precision = prec;
    if (!xyz) {
        if (fseek(r->fp, (long)padded, SEEK_CUR) != 0) return GORDER_XTC_ERR_FORMAT;
        return GORDER_XTC_OK;
    }
    r->buf.resize(padded + 8);
    if (!read_exact(r->fp, r->buf.data(), padded)) return GORDER_XTC_ERR_FORMAT;
    memset(r->buf.data() + padded, 0, 8);
    const int st = decode_ints(r, minint, maxint, smallidx, r->buf.data(), padded);
    if (st != GORDER_XTC_OK) return st;
    const float inv_precision = 1.0f / prec;   // GROMACS xdrfile: coordinate = int * (1 / precision)
    const int *q = r->ints.data();
    if (all) {
        for (size_t k = 0; k < (size_t)natoms * 3; k++) xyz[k] = (float)q[k] * inv_precision;
    } else {
        for (size_t k = 0; k < r->group.size(); k++) {
            const size_t a = r->group[k];
            xyz[3 * k + 0] = (float)q[3 * a + 0] * inv_precision;
            xyz[3 * k + 1] = (float)q[3 * a + 1] * inv_precision;
            xyz[3 * k + 2] = (float)q[3 * a + 2] * inv_precision;
        }
    }
    return GORDER_XTC_OK;
}

}  // extern "C"

// ---- writer ------------------------------------------------------------------------------------------
// The compression side of the same published format (xdr3dcoord): coordinates become integers
// (round-half-away of x * precision in f32), the frame stores their bounding box, and atoms go out either at full
// width (one mixed-radix number of `bitsize` bits) or, for runs of up to 8 neighbours closer than `smallnum` grid
// steps, as small offsets from their predecessor; the size of "small" adapts by one table step per atom group.
// Tooling only: the reference never writes trajectories.  It exists so that tests and the end-to-end benchmark can
// produce multi-frame XTC input for the reader -> GPU pipeline without any file the checkout does not hold.
namespace {
struct BitWriter {
    std::vector<uint8_t> out;
    uint32_t lastbyte = 0;
    int lastbits = 0;
    void bits(int nbits, uint32_t num) {          // MSB first
        while (nbits >= 8) {
            lastbyte = (lastbyte << 8) | ((num >> (nbits - 8)) & 0xffu);
            out.push_back((uint8_t)(lastbyte >> lastbits));
            nbits -= 8;
        }
        if (nbits > 0) {
            lastbyte = (lastbyte << nbits) | (num & ((1u << nbits) - 1u));
            lastbits += nbits;
            if (lastbits >= 8) {
                lastbits -= 8;
                out.push_back((uint8_t)(lastbyte >> lastbits));
            }
        }
    }
    // three integers as one mixed-radix number, least significant byte first, `nbits` bits in all
    void ints(int nbits, const uint32_t sizes[3], const uint32_t nums[3]) {
        unsigned __int128 v = nums[0];
        v = v * sizes[1] + nums[1];
        v = v * sizes[2] + nums[2];
        uint8_t bytes[16];
        int nbytes = 0;
        do { bytes[nbytes++] = (uint8_t)(v & 0xff); v >>= 8; } while (v != 0);
        if (nbits >= nbytes * 8) {
            for (int i = 0; i < nbytes; i++) bits(8, bytes[i]);
            int pad = nbits - nbytes * 8;
            while (pad > 0) { const int k = pad > 24 ? 24 : pad; bits(k, 0); pad -= k; }
        } else {
            for (int i = 0; i < nbytes - 1; i++) bits(8, bytes[i]);
            bits(nbits - (nbytes - 1) * 8, bytes[nbytes - 1]);
        }
    }
    void finish() {
        if (lastbits > 0) out.push_back((uint8_t)(lastbyte << (8 - lastbits)));
    }
};
void put32(std::vector<uint8_t> &o, uint32_t v) { o.push_back(v >> 24); o.push_back(v >> 16); o.push_back(v >> 8); o.push_back(v); }
void putf(std::vector<uint8_t> &o, float f) { uint32_t u; memcpy(&u, &f, 4); put32(o, u); }
}  // namespace

struct gorder_xtc_writer {
    FILE *fp = nullptr;
    uint32_t natoms = 0;
    float precision = 1000.0f;
    std::vector<int> ints;
    std::vector<uint8_t> head;
};

extern "C" {

int gorder_xtc_writer_open(const char *path, uint32_t n_atoms, float precision, gorder_xtc_writer **out) {
    if (!path || !out || n_atoms == 0 || !(precision > 0.0f)) return GORDER_XTC_ERR_ARGUMENT;
    *out = nullptr;
    FILE *fp = fopen(path, "wb");
    if (!fp) return GORDER_XTC_ERR_OPEN;
    gorder_xtc_writer *w = new gorder_xtc_writer();
    w->fp = fp;
    w->natoms = n_atoms;
    w->precision = precision;
    *out = w;
    return GORDER_XTC_OK;
}

void gorder_xtc_writer_close(gorder_xtc_writer *w) {
    if (!w) return;
    if (w->fp) fclose(w->fp);
    delete w;
}

int gorder_xtc_writer_add(gorder_xtc_writer *w, const float *xyz, const float *box9, int64_t step, float time_ps) {
    if (!w || !w->fp || !xyz || !box9) return GORDER_XTC_ERR_ARGUMENT;
    const uint32_t natoms = w->natoms;
    std::vector<uint8_t> &o = w->head;
    o.clear();
    put32(o, 1995u);
    put32(o, natoms);
    put32(o, (uint32_t)(int32_t)step);
    putf(o, time_ps);
    for (int k = 0; k < 9; k++) putf(o, box9[k]);
    put32(o, natoms);
    if (natoms <= 9) {
        for (size_t k = 0; k < (size_t)natoms * 3; k++) putf(o, xyz[k]);
        return fwrite(o.data(), 1, o.size(), w->fp) == o.size() ? GORDER_XTC_OK : GORDER_XTC_ERR_OPEN;
    }
    // ---- integers, their bounding box, the smallest step between consecutive atoms
    w->ints.resize((size_t)natoms * 3);
    int *ip = w->ints.data();
    int minint[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, maxint[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    int mindiff = INT32_MAX;
    int old[3] = {0, 0, 0};
    for (uint32_t a = 0; a < natoms; a++) {
        int cur[3];
        for (int c = 0; c < 3; c++) {
            const float x = xyz[3 * (size_t)a + c];
            const float lf = x >= 0.0f ? x * w->precision + 0.5f : x * w->precision - 0.5f;
            if (!(std::fabs(lf) < 2147483520.0f)) return GORDER_XTC_ERR_ARGUMENT;     // does not fit an int (or NaN)
            cur[c] = (int)lf;
            if (cur[c] < minint[c]) minint[c] = cur[c];
            if (cur[c] > maxint[c]) maxint[c] = cur[c];
            ip[3 * (size_t)a + c] = cur[c];
        }
        const int64_t diff = (int64_t)std::abs((int64_t)old[0] - cur[0]) + std::abs((int64_t)old[1] - cur[1]) +
                             std::abs((int64_t)old[2] - cur[2]);
        if (a > 0 && diff < mindiff) mindiff = (int)diff;
        old[0] = cur[0]; old[1] = cur[1]; old[2] = cur[2];
    }
    uint32_t sizeint[3];
    int bitsizeint[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        const int64_t sz = (int64_t)maxint[c] - (int64_t)minint[c] + 1;
        if (sz > 0xffffffffll) return GORDER_XTC_ERR_ARGUMENT;
        sizeint[c] = (uint32_t)sz;
    }
    int bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int c = 0; c < 3; c++) bitsizeint[c] = size_of_int(sizeint[c]);
        bitsize = 0;
    } else {
        bitsize = size_of_ints(sizeint);
    }
    int smallidx = kFirstIdx;
    while (smallidx < kLastIdx - 1 && kMagicInts[smallidx] < mindiff) smallidx++;
    putf(o, w->precision);
    for (int c = 0; c < 3; c++) put32(o, (uint32_t)minint[c]);
    for (int c = 0; c < 3; c++) put32(o, (uint32_t)maxint[c]);
    put32(o, (uint32_t)smallidx);
    const int maxidx = std::min(kLastIdx - 1, smallidx + 8), minidx = maxidx - 8;
    int smaller = kMagicInts[std::max(kFirstIdx, smallidx - 1)] / 2;
    int smallnum = kMagicInts[smallidx] / 2;
    uint32_t sizesmall[3] = {(uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx], (uint32_t)kMagicInts[smallidx]};
    const int larger = kMagicInts[maxidx] / 2;
    BitWriter bw;
    bw.out.reserve((size_t)natoms * 5);
    int prevcoord[3] = {0, 0, 0};
    int prevrun = -1;
    uint32_t i = 0;
    auto near = [](const int *p, const int *q, int lim) {
        return std::abs(p[0] - q[0]) < lim && std::abs(p[1] - q[1]) < lim && std::abs(p[2] - q[2]) < lim;
    };
    while (i < natoms) {
        int *thiscoord = ip + 3 * (size_t)i;
        bool is_small = false;
        int is_smaller;
        if (smallidx < maxidx && i >= 1 && near(thiscoord, prevcoord, larger)) is_smaller = 1;
        else if (smallidx > minidx) is_smaller = -1;
        else is_smaller = 0;
        if (i + 1 < natoms && near(thiscoord, thiscoord + 3, smallnum)) {
            // the first atom of a run is stored after the second (water: the oxygen between its hydrogens)
            for (int c = 0; c < 3; c++) std::swap(thiscoord[c], thiscoord[3 + c]);
            is_small = true;
        }
        const uint32_t big[3] = {(uint32_t)(thiscoord[0] - minint[0]), (uint32_t)(thiscoord[1] - minint[1]),
                                 (uint32_t)(thiscoord[2] - minint[2])};
        if (bitsize == 0) for (int c = 0; c < 3; c++) bw.bits(bitsizeint[c], big[c]);
        else bw.ints(bitsize, sizeint, big);
        for (int c = 0; c < 3; c++) prevcoord[c] = thiscoord[c];
        thiscoord += 3;
        i++;
        int run = 0;
        uint32_t small[24];
        if (!is_small && is_smaller == -1) is_smaller = 0;
        while (is_small && run < 8 * 3) {
            if (is_smaller == -1) {
                const int64_t d0 = thiscoord[0] - prevcoord[0], d1 = thiscoord[1] - prevcoord[1], d2 = thiscoord[2] - prevcoord[2];
                if (d0 * d0 + d1 * d1 + d2 * d2 >= (int64_t)smaller * smaller) is_smaller = 0;
            }
            for (int c = 0; c < 3; c++) small[run++] = (uint32_t)(thiscoord[c] - prevcoord[c] + smallnum);
            for (int c = 0; c < 3; c++) prevcoord[c] = thiscoord[c];
            i++;
            thiscoord += 3;
            is_small = i < natoms && near(thiscoord, prevcoord, smallnum);
        }
        if (run != prevrun || is_smaller != 0) {
            prevrun = run;
            bw.bits(1, 1);
            bw.bits(5, (uint32_t)(run + is_smaller + 1));
        } else {
            bw.bits(1, 0);
        }
        for (int k = 0; k < run; k += 3) bw.ints(smallidx, sizesmall, small + k);
        if (is_smaller != 0) {
            smallidx += is_smaller;
            if (is_smaller < 0) {
                smallnum = smaller;
                smaller = kMagicInts[smallidx - 1] / 2;
            } else {
                smaller = smallnum;
                smallnum = kMagicInts[smallidx] / 2;
            }
            sizesmall[0] = sizesmall[1] = sizesmall[2] = (uint32_t)kMagicInts[smallidx];
        }
    }
    bw.finish();
    put32(o, (uint32_t)bw.out.size());
    while (bw.out.size() % 4) bw.out.push_back(0);
    if (fwrite(o.data(), 1, o.size(), w->fp) != o.size()) return GORDER_XTC_ERR_OPEN;
    if (fwrite(bw.out.data(), 1, bw.out.size(), w->fp) != bw.out.size()) return GORDER_XTC_ERR_OPEN;
    return GORDER_XTC_OK;
}

}  // extern "C"

extern "C" {

int64_t gorder_xtc_read_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                               double *last_time, float *xyz, float *box9, float *time_ps, uint64_t capacity) {
    if (!r || !state || !last_time || !xyz || !box9 || step == 0) return GORDER_XTC_ERR_ARGUMENT;
    const size_t nout = gorder_xtc_n_atoms_out(r);
    uint64_t written = 0;
    while (written < capacity) {
        // peek the header to decide whether the coordinates are needed at all
        const long pos = ftell(r->fp);
        float t = 0.0f;
        int st = gorder_xtc_next(r, nullptr, nullptr, nullptr, &t, nullptr);
        if (st == GORDER_XTC_EOF) break;
        if (st != GORDER_XTC_OK) return st;
        if ((double)t == *last_time) continue;             // duplicate frame at a file boundary
        *last_time = (double)t;
        if (t < begin_ps) continue;
        if (end_ps >= 0.0f && t > end_ps) break;
        const uint64_t k = (*state)++;
        if (k % step != 0) continue;
        if (fseek(r->fp, pos, SEEK_SET) != 0) return GORDER_XTC_ERR_FORMAT;
        st = gorder_xtc_next(r, xyz + written * nout * 3, box9 + written * 9, nullptr, &t, nullptr);
        if (st != GORDER_XTC_OK) return st;
        if (time_ps) time_ps[written] = t;
        written++;
    }
    return (int64_t)written;
}

int64_t gorder_xtc_skip_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                               double *last_time, uint64_t max_frames) {
    if (!r || !state || !last_time || step == 0) return GORDER_XTC_ERR_ARGUMENT;
    uint64_t passed = 0;
    while (passed < max_frames) {
        float t = 0.0f;
        const int st = gorder_xtc_next(r, nullptr, nullptr, nullptr, &t, nullptr);
        if (st == GORDER_XTC_EOF) break;
        if (st != GORDER_XTC_OK) return st;
        if ((double)t == *last_time) continue;             // duplicate frame at a file boundary
        *last_time = (double)t;
        if (t < begin_ps) continue;
        if (end_ps >= 0.0f && t > end_ps) { fseek(r->fp, 0, SEEK_END); break; }
        const uint64_t k = (*state)++;
        if (k % step != 0) continue;
        passed++;
    }
    return (int64_t)passed;
}

int64_t gorder_xtc_read_window_mt(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step,
                                  uint64_t *state, double *last_time, float *xyz, float *box9, float *time_ps,
                                  uint64_t capacity, uint32_t n_threads) {
    if (!r || !state || !last_time || !xyz || !box9 || step == 0) return GORDER_XTC_ERR_ARGUMENT;
    if (n_threads <= 1) return gorder_xtc_read_window(r, begin_ps, end_ps, step, state, last_time, xyz, box9, time_ps, capacity);
    // pass 1 (sequential, headers only): which frames, and where they start
    std::vector<long> offsets;
    while (offsets.size() < capacity) {
        const long pos = ftell(r->fp);
        float t = 0.0f;
        const int st = gorder_xtc_next(r, nullptr, nullptr, nullptr, &t, nullptr);
        if (st == GORDER_XTC_EOF) break;
        if (st != GORDER_XTC_OK) return st;
        if ((double)t == *last_time) continue;             // duplicate frame at a file boundary
        *last_time = (double)t;
        if (t < begin_ps) continue;
        if (end_ps >= 0.0f && t > end_ps) { fseek(r->fp, 0, SEEK_END); break; }
        const uint64_t k = (*state)++;
        if (k % step != 0) continue;
        offsets.push_back(pos);
    }
    // pass 2: every worker decodes its share through a reader of its own
    const size_t nout = gorder_xtc_n_atoms_out(r), n = offsets.size();
    const uint32_t nt = (uint32_t)std::min<size_t>(n_threads, std::max<size_t>(n, 1));
    std::vector<int> status(nt, GORDER_XTC_OK);
    std::vector<std::thread> pool;
    for (uint32_t w = 0; w < nt; w++)
        pool.emplace_back([&, w]() {
            gorder_xtc_reader *mine = nullptr;
            int st = gorder_xtc_open(r->path.c_str(), r->group.empty() ? nullptr : r->group.data(), (uint32_t)r->group.size(), &mine);
            for (size_t i = w; st == GORDER_XTC_OK && i < n; i += nt) {
                if (fseek(mine->fp, offsets[i], SEEK_SET) != 0) { st = GORDER_XTC_ERR_FORMAT; break; }
                float t = 0.0f;
                st = gorder_xtc_next(mine, xyz + i * nout * 3, box9 + i * 9, nullptr, &t, nullptr);
                if (st == GORDER_XTC_OK && time_ps) time_ps[i] = t;
            }
            gorder_xtc_close(mine);
            status[w] = st;
        });
    for (auto &th : pool) th.join();
    for (int st : status)
        if (st != GORDER_XTC_OK) return st == GORDER_XTC_EOF ? GORDER_XTC_ERR_FORMAT : st;
    return (int64_t)n;
}

}  // extern "C"

// ---- packing for the device decoder ------------------------------------------------------------------
// A pool of copying threads that outlives one window: gorder_xtc_pack_window_pool hands the block copies of a window
// to it and returns as soon as the headers are scanned, so that the (sequential) header scan of the next file or window
// runs while the blocks of this one are still being copied (scan 0.28 ms + copy 0.83 ms per 500 V-AA frames, measured).
struct gorder_xtc_pool {
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<int()>> jobs;
    size_t running = 0;
    int status = GORDER_XTC_OK;      // first failure since the last wait
    bool stop = false;
    void loop() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                job = std::move(jobs.front());
                jobs.pop_front();
                running++;
            }
            const int st = job();
            std::lock_guard<std::mutex> lk(mu);
            if (st != GORDER_XTC_OK && status == GORDER_XTC_OK) status = st;
            running--;
            if (jobs.empty() && running == 0) cv_idle.notify_all();
        }
    }
};

extern "C" {

int gorder_xtc_pool_create(uint32_t n_threads, gorder_xtc_pool **out) {
    if (!out) return GORDER_XTC_ERR_ARGUMENT;
    gorder_xtc_pool *p = new gorder_xtc_pool();
    for (uint32_t w = 0; w < std::max(1u, n_threads); w++) p->workers.emplace_back([p] { p->loop(); });
    *out = p;
    return GORDER_XTC_OK;
}
int gorder_xtc_pool_wait(gorder_xtc_pool *p) {
    if (!p) return GORDER_XTC_ERR_ARGUMENT;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_idle.wait(lk, [&] { return p->jobs.empty() && p->running == 0; });
    const int st = p->status;
    p->status = GORDER_XTC_OK;
    return st;
}
void gorder_xtc_pool_destroy(gorder_xtc_pool *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    for (auto &t : p->workers) t.join();
    delete p;
}


int gorder_xtc_probe(const char *path, uint32_t *n_atoms, uint64_t *file_bytes, uint32_t *first_frame_bytes) {
    if (!path) return GORDER_XTC_ERR_ARGUMENT;
    FILE *fp = fopen(path, "rb");
    if (!fp) return GORDER_XTC_ERR_OPEN;
    uint8_t head[92];
    const size_t got = fread(head, 1, sizeof(head), fp);
    if (file_bytes) {
        *file_bytes = 0;
        if (fseeko(fp, 0, SEEK_END) == 0) *file_bytes = (uint64_t)ftello(fp);
    }
    fclose(fp);
    if (got < 8) return GORDER_XTC_ERR_FORMAT;
    if (be32(head) != 1995u) return 0;
    const uint32_t na = be32(head + 4);
    if (n_atoms) *n_atoms = na;
    if (first_frame_bytes) {       // header + coordinate block of the first frame (0: the file is too short to tell)
        *first_frame_bytes = 0;
        if (na <= 9 && got >= 56) *first_frame_bytes = 56u + 12u * na;
        else if (na > 9 && got == sizeof(head)) *first_frame_bytes = 92u + ((be32(head + 88) + 3u) & ~3u);
    }
    return 1;
}
int gorder_xtc_is_xtc(const gorder_xtc_reader *r) { return (r && !r->trr && !r->gro) ? 1 : 0; }
uint32_t gorder_xtc_n_atoms_needed(const gorder_xtc_reader *r) {
    return r ? (r->n_needed ? std::min(r->n_needed, r->natoms) : r->natoms) : 0;
}

}  // extern "C"

namespace {
struct PackSrc { off_t pos; uint32_t n; uint64_t dst; };
// n bytes to a 32-byte aligned destination with non-temporal stores (the blob is written once and read by the DMA engine)
__attribute__((target("avx2"))) void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    if (i < n) memcpy(dst + i, src + i, n - i);
    _mm_sfence();
}
bool have_avx2() {
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
// the blocks [i0, i1) of a window into the blob: from the file's mapping when there is one (streaming stores), else by
// pread (which does not move the file position)
int pack_copy(int fd, const FileMap *map, const PackSrc *src, size_t i0, size_t i1, uint8_t *blob) {
    const bool stream = map && map->base && have_avx2() && !getenv("GORDER_XTC_PREAD");
    for (size_t i = i0; i < i1; i++) {
        uint8_t *dst = blob + src[i].dst;
        size_t done = 0;
        if (stream && ((uintptr_t)dst & 31u) == 0 && (size_t)src[i].pos + src[i].n <= map->size) {
            // (Measured in round 4 and not kept: MADV_POPULATE_READ on the block's pages before the copy — one call instead
            // of 31 minor faults per block: 217 k against 222 k frames/s on 8 GB of distinct frames; the copy out of DRAM
            // itself, 27 GB/s on the 16 CPUs of the box's quota, is the bound, not the faults.)
            stream_copy_avx2(dst, map->base + src[i].pos, src[i].n);
            done = src[i].n;
        }
        while (done < src[i].n) {
            const ssize_t g = pread(fd, dst + done, src[i].n - done, src[i].pos + (off_t)done);
            if (g <= 0) return GORDER_XTC_ERR_FORMAT;
            done += (size_t)g;
        }
        const size_t end = (size_t)((((uint64_t)src[i].n + 63u) & ~63ull) + 64u);
        memset(dst + src[i].n, 0, end - src[i].n);
    }
    return GORDER_XTC_OK;
}

int64_t pack_window_impl(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                         double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                         gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                         uint32_t n_threads, gorder_xtc_pool *pool, uint32_t prefix_q16 = 65536u, int64_t *file_pos = nullptr) {
    if (!r || !r->fp || r->trr || r->gro || !state || !last_time || !blob || !blob_bytes || !frames || !box9 || step == 0)
        return GORDER_XTC_ERR_ARGUMENT;
    *blob_bytes = 0;
    const auto t_scan0 = std::chrono::steady_clock::now();
    // pass 1 (sequential): the headers — which frames, where their blocks lie, what the decoder needs to know.
    // One pread of 92 bytes per frame, the position kept here (the FILE is moved once, at the end).
    std::vector<PackSrc> src;
    uint64_t used = 0;
    const uint32_t natoms = r->natoms;
    const int fd = fileno(r->fp);
    off_t pos = (off_t)ftello(r->fp);
    bool to_end = false;
    const uint64_t state_at_entry = *state;        // a call that returns NO_SPACE leaves reader AND selection state untouched
    const double last_time_at_entry = *last_time;
    struct stat sb;
    const off_t file_end = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) ? sb.st_size : (off_t)-1;
    if (!r->map_tried) {            // the file mapped once, for the block copies and the header scan
        r->map_tried = true;
        if (file_end > 0) {
            void *m = mmap(nullptr, (size_t)file_end, PROT_READ, MAP_SHARED, fd, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, (size_t)file_end, MADV_SEQUENTIAL);
                r->map = std::make_shared<FileMap>();
                r->map->base = static_cast<const uint8_t *>(m);
                r->map->size = (size_t)file_end;
            }
        }
    }
    // A file that has SHRUNK since it was mapped must not be touched through the mapping any more (pages behind the new
    // end raise SIGBUS): this reader goes back to pread for good.  A file that has GROWN keeps its mapping for the part
    // that was there and takes pread behind it.  (A file truncated while a copy is in flight can still fault: for
    // trajectories another process rewrites, or on file systems that return I/O errors late, set GORDER_XTC_PREAD=1.)
    if (r->map && (getenv("GORDER_XTC_PREAD") || (file_end >= 0 && (size_t)file_end < r->map->size))) r->map.reset();
    const FileMap *map = r->map.get();
    while (src.size() < capacity) {
        const off_t pos0 = pos;
        uint8_t head[56 + 36];
        ssize_t got;
        if (map && map->base && pos0 >= 0 && (size_t)pos0 + sizeof(head) <= map->size) {      // (no system call per frame)
            got = (ssize_t)sizeof(head);
            memcpy(head, map->base + pos0, sizeof(head));
        } else {            // the last bytes of the mapping, or behind it (the file has grown): the file itself decides
            got = pread(fd, head, sizeof(head), pos0);
        }
        if (got == 0) break;
        if (got < 56 || be32(head) != 1995u || be32(head + 4) != natoms || be32(head + 52) != natoms)
            return GORDER_XTC_ERR_FORMAT;
        const float t = bef(head + 12);
        gorder_xtc_frame_t fr{};
        uint32_t block = 0;       // bytes of the coordinate block in the file
        off_t pos_block;
        if (natoms <= 9) {
            fr.kind = 1;
            fr.n_bytes = block = natoms * 12u;
            pos_block = pos0 + 56;
        } else {
            if (got != (ssize_t)sizeof(head)) return GORDER_XTC_ERR_FORMAT;
            const uint8_t *h2 = head + 56;
            for (int k = 0; k < 3; k++) {
                const int minint = (int32_t)be32(h2 + 4 + 4 * k), maxint = (int32_t)be32(h2 + 16 + 4 * k);
                const int64_t sz = (int64_t)maxint - (int64_t)minint + 1;
                if (sz <= 0 || sz > 0xffffffffll) return GORDER_XTC_ERR_FORMAT;
                fr.minint[k] = minint;
                fr.sizeint[k] = (uint32_t)sz;
            }
            if ((fr.sizeint[0] | fr.sizeint[1] | fr.sizeint[2]) > 0xffffff) {
                fr.bitsize = 0;
                for (int k = 0; k < 3; k++) fr.bitsizeint |= (uint32_t)size_of_int(fr.sizeint[k]) << (8 * k);
            } else {
                fr.bitsize = (uint32_t)size_of_ints(fr.sizeint);
            }
            fr.recip1 = reciprocal_of(fr.sizeint[1]);
            fr.recip2 = reciprocal_of(fr.sizeint[2]);
            fr.smallidx = (int32_t)be32(h2 + 28);
            if (fr.smallidx < kFirstIdx || fr.smallidx >= kLastIdx) return GORDER_XTC_ERR_FORMAT;
            fr.inv_precision = 1.0f / bef(h2);
            block = (uint32_t)(((size_t)be32(h2 + 32) + 3) & ~(size_t)3);
            fr.n_bytes = block;
            pos_block = pos0 + (off_t)sizeof(head);
        }
        // only the leading part of the block (the analysed atoms come first in the frame and the decoder stops behind
        // them): `copy` bytes are copied and described, bit 1 of `kind` says that the block goes on in the file
        uint32_t copy = block;
        if (prefix_q16 < 65536u && natoms > 9) {
            const uint64_t c = (((uint64_t)block * prefix_q16 >> 16) + 2048u + 3u) & ~3ull;
            if (c < block) { copy = (uint32_t)c; fr.kind |= 2u; fr.n_bytes = copy; }
        }
        pos = pos_block + (off_t)block;
        if (file_end >= 0 && pos > file_end) return GORDER_XTC_ERR_FORMAT;     // a byte count the file cannot hold
        // selection: as in gorder_xtc_read_window
        if ((double)t == *last_time) continue;             // duplicate frame at a file boundary
        const double time_before = *last_time;
        *last_time = (double)t;
        if (t < begin_ps) continue;
        if (end_ps >= 0.0f && t > end_ps) { to_end = true; break; }
        const uint64_t k = (*state)++;
        if (k % step != 0) continue;
        const uint64_t need = (((uint64_t)copy + 63u) & ~63ull) + 64u;      // whole 64-byte pieces + one piece of zeros
        if (used + need > blob_capacity) {                 // does not fit any more: this frame opens the next window
            if (src.empty()) {                             // not even one frame: the FILE is still where it was at entry,
                *state = state_at_entry;                   // so the frames passed over so far (stepped over, the duplicate at
                *last_time = last_time_at_entry;           // a file boundary) must not be counted either
                return GORDER_XTC_ERR_NO_SPACE;
            }
            (*state)--;                                    // (nothing of it has happened: a later call meets it again)
            *last_time = time_before;
            pos = pos0;
            break;
        }
        const size_t i = src.size();
        fr.offset = used;
        used += need;
        frames[i] = fr;
        for (int q = 0; q < 9; q++) box9[9 * i + q] = bef(head + 16 + 4 * q);
        if (time_ps) time_ps[i] = t;
        if (file_pos) file_pos[i] = (int64_t)pos0;
        src.push_back({pos_block, copy, fr.offset});
    }
    if (to_end ? fseek(r->fp, 0, SEEK_END) != 0 : fseeko(r->fp, pos, SEEK_SET) != 0) return GORDER_XTC_ERR_FORMAT;
    // pass 2: the blocks, by several readers at once
    const size_t n = src.size();
    if (n == 0) return 0;
    const auto t_copy0 = std::chrono::steady_clock::now();
    if (pool) {
        // asynchronously: contiguous shares of the window, one job per worker; the descriptor is duplicated so that
        // the reader may be closed before the copies are done
        const size_t nj = std::min<size_t>(pool->workers.size(), n);
        auto shared = std::make_shared<std::vector<PackSrc>>(std::move(src));
        const int fd2 = dup(fd);
        if (fd2 < 0) return GORDER_XTC_ERR_OPEN;
        auto left = std::make_shared<std::atomic<size_t>>(nj);
        std::shared_ptr<FileMap> keep = r->map;
        {
            std::lock_guard<std::mutex> lk(pool->mu);
            for (size_t w = 0; w < nj; w++) {
                const size_t i0 = n * w / nj, i1 = n * (w + 1) / nj;
                pool->jobs.emplace_back([shared, fd2, left, i0, i1, blob, keep]() {
                    const int st = pack_copy(fd2, keep.get(), shared->data(), i0, i1, blob);
                    if (left->fetch_sub(1) == 1) close(fd2);
                    return st;
                });
            }
        }
        pool->cv_work.notify_all();
        *blob_bytes = used;
        return (int64_t)n;
    }
    const uint32_t nt = (uint32_t)std::min<size_t>(std::max(1u, n_threads), n);
    std::vector<int> status(nt, GORDER_XTC_OK);
    auto work = [&](uint32_t w) { status[w] = pack_copy(fd, map, src.data(), n * w / nt, n * (w + 1) / nt, blob); };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> threads;
        for (uint32_t w = 0; w < nt; w++) threads.emplace_back(work, w);
        for (auto &th : threads) th.join();
    }
    for (int st : status)
        if (st != GORDER_XTC_OK) return st;
    if (getenv("GORDER_XTC_PACK_TIMING")) {     // development aid: where a window's time goes
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "pack_window: %zu frames, scan %.3f ms, copy %.3f ms (%u threads, %.1f MB)\n", n,
                std::chrono::duration<double, std::milli>(t_copy0 - t_scan0).count(),
                std::chrono::duration<double, std::milli>(t1 - t_copy0).count(), nt, used / 1e6);
    }
    *blob_bytes = used;
    return (int64_t)n;
}
}  // namespace

extern "C" {

int64_t gorder_xtc_pack_window(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                               double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                               gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                               uint32_t n_threads) {
    return pack_window_impl(r, begin_ps, end_ps, step, state, last_time, blob, blob_capacity, blob_bytes, frames, box9,
                            time_ps, capacity, n_threads, nullptr);
}

int64_t gorder_xtc_pack_window_ex(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                                  double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                                  gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                                  uint32_t n_threads, gorder_xtc_pool *pool, uint32_t prefix_q16, int64_t *file_pos) {
    return pack_window_impl(r, begin_ps, end_ps, step, state, last_time, blob, blob_capacity, blob_bytes, frames, box9,
                            time_ps, capacity, n_threads ? n_threads : 1, pool, prefix_q16 ? prefix_q16 : 65536u, file_pos);
}

int gorder_xtc_read_at(gorder_xtc_reader *r, int64_t file_pos, float *xyz, float *box9) {
    if (!r || !r->fp || !xyz) return GORDER_XTC_ERR_ARGUMENT;
    if (fseeko(r->fp, (off_t)file_pos, SEEK_SET) != 0) return GORDER_XTC_ERR_FORMAT;
    return gorder_xtc_next(r, xyz, box9, nullptr, nullptr, nullptr);
}

int64_t gorder_xtc_pack_window_pool(gorder_xtc_reader *r, float begin_ps, float end_ps, uint32_t step, uint64_t *state,
                                    double *last_time, uint8_t *blob, uint64_t blob_capacity, uint64_t *blob_bytes,
                                    gorder_xtc_frame_t *frames, float *box9, float *time_ps, uint64_t capacity,
                                    gorder_xtc_pool *pool) {
    if (!pool) return GORDER_XTC_ERR_ARGUMENT;
    return pack_window_impl(r, begin_ps, end_ps, step, state, last_time, blob, blob_capacity, blob_bytes, frames, box9,
                            time_ps, capacity, 1, pool);
}

}  // extern "C"
