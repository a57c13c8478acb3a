// The host reader (xtc_reader.cpp) under AddressSanitizer / UBSan / ThreadSanitizer: every entry point of
// include/gorder_xtc.h on three of the reference's files — windowed multi-threaded reading, header-only counting,
// packing for the device decoder (synchronous and through the copy pool, with windows that split), the writer.
// Built and run by tests/test_reader_sanitizers_cpu.py (CPU only; GOLDEN and OUTFILE come from the command line).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "gorder_xtc.h"
int main() {
    const char *files[] = {GOLDEN "/multiple_resid_same_name.xtc", GOLDEN "/cg3.xtc", GOLDEN "/pcpepg4.xtc"};
    for (const char *path : files) {
        uint32_t na = 0;
        uint64_t fbytes = 0; uint32_t first = 0;
        if (gorder_xtc_probe(path, &na, &fbytes, &first) != 1 || first == 0 || first > fbytes) return 1;
        std::vector<uint32_t> group;
        for (uint32_t i = 0; i < na; i += 3) group.push_back(i);
        for (int g = 0; g < 2; g++) {
            gorder_xtc_reader *r = nullptr;
            if (gorder_xtc_open(path, g ? group.data() : nullptr, g ? (uint32_t)group.size() : 0, &r)) return 2;
            const uint32_t nout = gorder_xtc_n_atoms_out(r);
            uint64_t state = 0; double last = -INFINITY;
            std::vector<float> xyz((size_t)7 * nout * 3), box(7 * 9), t(7);
            int64_t total = 0, got;
            while ((got = gorder_xtc_read_window_mt(r, 0, -1, 2, &state, &last, xyz.data(), box.data(), t.data(), 7, 3)) > 0) total += got;
            if (got < 0) return 3;
            gorder_xtc_close(r);
            // count
            gorder_xtc_open(path, nullptr, 0, &r);
            state = 0; last = -INFINITY;
            const int64_t cnt = gorder_xtc_skip_window(r, 0, -1, 2, &state, &last, UINT64_MAX);
            gorder_xtc_close(r);
            if (cnt != total) { printf("count %lld vs %lld\n", (long long)cnt, (long long)total); return 4; }
            // pack: sync and pooled, small blob so that windows split
            for (int pooled = 0; pooled < 2; pooled++) {
                gorder_xtc_open(path, g ? group.data() : nullptr, g ? (uint32_t)group.size() : 0, &r);
                gorder_xtc_pool *pool = nullptr;
                if (pooled) gorder_xtc_pool_create(4, &pool);
                state = 0; last = -INFINITY;
                const size_t cap = (size_t)na * 16 + 8192;
                std::vector<uint8_t> blob(cap);
                std::vector<gorder_xtc_frame_t> fr(5);
                int64_t packed = 0;
                for (;;) {
                    uint64_t used = 0;
                    got = pooled ? gorder_xtc_pack_window_pool(r, 0, -1, 2, &state, &last, blob.data(), cap, &used, fr.data(), box.data(), t.data(), 5, pool)
                                 : gorder_xtc_pack_window(r, 0, -1, 2, &state, &last, blob.data(), cap, &used, fr.data(), box.data(), t.data(), 5, 3);
                    if (pooled && got >= 0 && gorder_xtc_pool_wait(pool)) return 5;
                    if (got <= 0) break;
                    packed += got;
                }
                if (got < 0) { printf("pack error %lld\n", (long long)got); return 6; }
                if (packed != total) { printf("packed %lld vs %lld\n", (long long)packed, (long long)total); return 7; }
                gorder_xtc_close(r);
                if (pool) gorder_xtc_pool_destroy(pool);
            }
        }
        // writer round trip
        gorder_xtc_reader *r = nullptr; gorder_xtc_open(path, nullptr, 0, &r);
        std::vector<float> x((size_t)na * 3), b(9); float tt, prec; int64_t st;
        if (gorder_xtc_next(r, x.data(), b.data(), &st, &tt, &prec)) return 8;
        gorder_xtc_close(r);
        gorder_xtc_writer *w = nullptr;
        if (gorder_xtc_writer_open(OUTFILE, na, prec > 0 ? prec : 1000.0f, &w)) return 9;
        if (gorder_xtc_writer_add(w, x.data(), b.data(), st, tt)) return 10;
        gorder_xtc_writer_close(w);
        printf("%s ok (%u atoms)\n", path, na);
    }
    return 0;
}
