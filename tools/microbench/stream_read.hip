// Read-only HBM streaming ceiling on this device, in the access shapes the order kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip && ./stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// (1) plain grid-stride float4 sum
template <bool NT>
__global__ __launch_bounds__(256) void k_linear(const v4f* __restrict__ p, size_t n4, float* out) {
    v4f acc = {0,0,0,0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        v4f a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride);
                  c = __builtin_nontemporal_load(p + i + 2*stride); d = __builtin_nontemporal_load(p + i + 3*stride); }
        else { a = p[i]; b = p[i + stride]; c = p[i + 2*stride]; d = p[i + 3*stride]; }
        acc += (a + b) + (c + d);
    }
    for (; i < n4; i += stride) acc += p[i];
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}
// (2) the order kernel's shape: block (tile, chunk) reads `win4` float4 per frame at stride `frame4`, G frames per step
template <int G, bool NT>
__global__ __launch_bounds__(256) void k_tiles(const v4f* __restrict__ p, uint32_t n_tiles, uint32_t win4, size_t frame4,
                                               uint32_t n_frames, uint32_t fpc, float* out) {
    const uint32_t tile = blockIdx.x % n_tiles, chunk = blockIdx.x / n_tiles;
    const uint32_t f0 = chunk * fpc, f1 = min(n_frames, f0 + fpc);
    constexpr uint32_t TPF = 256 / G;
    const uint32_t sk = threadIdx.x / TPF, si = threadIdx.x % TPF;
    v4f acc = {0,0,0,0};
    for (uint32_t f = f0 + sk; f < f1; f += G) {
        const v4f* src = p + (size_t)f * frame4 + (size_t)tile * win4;
        for (uint32_t i = si; i < win4; i += TPF) acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t frame_floats = 25088ull * 3, n_frames = 10000;
    const size_t bytes = frame_floats * 4 * n_frames;
    v4f* d; float* out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        printf("%-44s %.4f ms  %.0f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    };
    const size_t n4 = bytes / 16;
    for (int blocks : {2048, 8192, 32768}) {
        char nm[96];
        snprintf(nm, 96, "linear float4, %d blocks", blocks);
        time(nm, [&] { hipLaunchKernelGGL(k_linear<false>, dim3(blocks), dim3(256), 0, 0, d, n4, out); });
        snprintf(nm, 96, "linear float4 nt, %d blocks", blocks);
        time(nm, [&] { hipLaunchKernelGGL(k_linear<true>, dim3(blocks), dim3(256), 0, 0, d, n4, out); });
    }
    const uint32_t n_tiles = 64, win4 = (uint32_t)(frame_floats / 4 / n_tiles);   // 294 float4 = 4704 B per tile
    for (uint32_t chunks : {16u, 48u, 192u, 768u}) {
        uint32_t fpc = (uint32_t)((n_frames + chunks - 1) / chunks); fpc = (fpc + 3) / 4 * 4;
        uint32_t nch = (uint32_t)((n_frames + fpc - 1) / fpc);
        char nm[96];
        snprintf(nm, 96, "tiles G=4, %u WGs", n_tiles * nch);
        time(nm, [&] { hipLaunchKernelGGL((k_tiles<4, false>), dim3(n_tiles * nch), dim3(256), 0, 0, d, n_tiles, win4, frame_floats / 4, (uint32_t)n_frames, fpc, out); });
        snprintf(nm, 96, "tiles G=4 nt, %u WGs", n_tiles * nch);
        time(nm, [&] { hipLaunchKernelGGL((k_tiles<4, true>), dim3(n_tiles * nch), dim3(256), 0, 0, d, n_tiles, win4, frame_floats / 4, (uint32_t)n_frames, fpc, out); });
        snprintf(nm, 96, "tiles G=1, %u WGs", n_tiles * nch);
        time(nm, [&] { hipLaunchKernelGGL((k_tiles<1, false>), dim3(n_tiles * nch), dim3(256), 0, 0, d, n_tiles, win4, frame_floats / 4, (uint32_t)n_frames, fpc, out); });
    }
    {   // CG-1M shape: 1 000 008 atoms/frame (12 MB), 3581 tiles of ~3.4 KB, 250 frames
        const size_t ff = 1000008ull * 3; const uint32_t nf = 250, nt = 3581, w4 = (uint32_t)(ff / 4 / nt);
        const size_t by = ff * 4 * nf;
        printf("CG-1M shape: %u tiles x %u float4, frame stride %.1f MB, %.2f GB\n", nt, w4, ff * 4 / 1e6, by / 1e9);
        for (uint32_t chunks : {1u, 3u, 10u}) {
            uint32_t fpc = (nf + chunks - 1) / chunks; fpc = (fpc + 3) / 4 * 4; uint32_t nch = (nf + fpc - 1) / fpc;
            hipLaunchKernelGGL((k_tiles<4, true>), dim3(nt * nch), dim3(256), 0, 0, d, nt, w4, ff / 4, nf, fpc, out); CK(hipDeviceSynchronize());
            float best = 1e9;
            for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_tiles<4, true>), dim3(nt * nch), dim3(256), 0, 0, d, nt, w4, ff / 4, nf, fpc, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
            printf("  tiles G=4 nt, %u WGs: %.4f ms %.0f GB/s\n", nt * nch, best, (double)nt * w4 * 16 * nf / (best * 1e-3) / 1e9);
        }
        // wider tiles: 4 adjacent tiles per WG pass (13.6 KB contiguous per frame)
        for (uint32_t chunks : {3u, 10u}) {
            const uint32_t nt4 = nt / 4, w44 = w4 * 4;
            uint32_t fpc = (nf + chunks - 1) / chunks; fpc = (fpc + 3) / 4 * 4; uint32_t nch = (nf + fpc - 1) / fpc;
            hipLaunchKernelGGL((k_tiles<4, true>), dim3(nt4 * nch), dim3(256), 0, 0, d, nt4, w44, ff / 4, nf, fpc, out); CK(hipDeviceSynchronize());
            float best = 1e9;
            for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_tiles<4, true>), dim3(nt4 * nch), dim3(256), 0, 0, d, nt4, w44, ff / 4, nf, fpc, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
            printf("  4x wider tiles, %u WGs: %.4f ms %.0f GB/s\n", nt4 * nch, best, (double)nt4 * w44 * 16 * nf / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
