// copy_bench.cpp — how fast can N threads move a file from the page cache into a (staging) buffer?
//   g++ -O2 -mavx2 -pthread -o copy_bench copy_bench.cpp && ./copy_bench FILE THREADS [piece_bytes]
// Three ways, the same pieces (what gorder_xtc_pack_window copies: one compressed block per frame):
//   pread        the kernel copies (copy_to_user) into the destination
//   mmap+memcpy  the file mapped once, glibc memcpy
//   mmap+stream  the file mapped once, non-temporal 32-byte stores (no read-for-ownership of the destination lines)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static void stream_copy(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    while (i < n && ((uintptr_t)(dst + i) & 31u)) { dst[i] = src[i]; i++; }
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a);
        _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c);
        _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    for (; i < n; i++) dst[i] = src[i];
    _mm_sfence();
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    const int nt = atoi(argv[2]);
    const size_t piece = argc > 3 ? (size_t)atol(argv[3]) : 125000;
    const int fd = open(argv[1], O_RDONLY);
    struct stat sb;
    fstat(fd, &sb);
    const size_t size = (size_t)sb.st_size, n_piece = size / piece;
    uint8_t *dst = (uint8_t *)aligned_alloc(4096, n_piece * piece + 4096);
    memset(dst, 1, n_piece * piece);                       // touched: no first-touch faults in the timed part
    const uint8_t *map = (const uint8_t *)mmap(nullptr, size, PROT_READ, MAP_SHARED, fd, 0);
    madvise((void *)map, size, MADV_WILLNEED);
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    for (size_t i = n_piece * t / nt; i < n_piece * (t + 1) / nt; i++) {
                        uint8_t *d = dst + i * piece;
                        if (mode == 0) { size_t done = 0; while (done < piece) done += (size_t)pread(fd, d + done, piece - done, (off_t)(i * piece + done)); }
                        else if (mode == 1) memcpy(d, map + i * piece, piece);
                        else stream_copy(d, map + i * piece, piece);
                    }
                });
            for (auto &x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%-12s %2d threads  %6.2f GB/s  (%.2f GB/s per thread)\n", mode == 0 ? "pread" : (mode == 1 ? "mmap+memcpy" : "mmap+stream"), nt,
                            n_piece * piece / s / 1e9, n_piece * piece / s / 1e9 / nt);
        }
    return 0;
}
