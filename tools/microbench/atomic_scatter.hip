// Microbenchmark: throughput of scattered 64-bit atomic adds on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scatter atomic_scatter.hip && ./atomic_scatter
// Compares device-scope (coherent across XCDs) with workgroup-scope (executed in the XCD's own L2;
// only correct with one replica per XCD) and ds_add_u64 into LDS, over table sizes from L2-resident
// to HBM-resident.  Used to choose the ordermap scatter scheme (DESIGN.md "Ordermaps").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int SCOPE, bool PER_XCD>
__global__ void k_scatter(unsigned long long *tab, uint32_t n_cells, uint32_t per_thread, uint32_t locality) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long *t = tab + (PER_XCD ? (size_t)xcc_id() * n_cells : 0);
    // `locality`: consecutive adds of one thread stay within a window of that many cells
    uint32_t base = (s >> 7) % n_cells;
    for (uint32_t i = 0; i < per_thread; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t c = locality ? (base + (s >> 9) % locality) % n_cells : (s >> 5) % n_cells;
        if (locality >= 1000) {   // lanes of a wave hit `locality-1000` consecutive cells per random base (lane clustering)
            const uint32_t g = locality - 1000, lane = threadIdx.x & 63;
            uint32_t sw = __shfl(s, (lane / g) * g, 64);
            c = ((sw >> 5) % (n_cells - 64)) + lane % g;
        }
        if constexpr (SCOPE == 99) __builtin_nontemporal_store((unsigned long long)s, &t[c]);   // plain scattered 8-byte store
        else __hip_atomic_fetch_add(&t[c], (unsigned long long)(s & 0xff), __ATOMIC_RELAXED, SCOPE);
    }
}

int main() {
    const uint32_t per_thread = 256, blocks = 8192, threads = 256;
    const double n_ops = (double)per_thread * blocks * threads;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t mb : {12u, 384u}) {
        const uint32_t n_cells = mb * 1024u * 1024u / 8u;
        unsigned long long *tab; hipMalloc(&tab, (size_t)n_cells * 8 * 8); hipMemset(tab, 0, (size_t)n_cells * 8 * 8);
        for (uint32_t loc : {0u, 1002u, 1004u, 1008u, 1016u, 1064u}) {
            float ms[3];
            for (int v = 0; v < 3; v++) {
                for (int rep = 0; rep < 3; rep++) {
                    hipEventRecord(e0);
                    if (v == 0) k_scatter<__HIP_MEMORY_SCOPE_AGENT, false><<<blocks, threads>>>(tab, n_cells, per_thread, loc);
                    if (v == 1) k_scatter<__HIP_MEMORY_SCOPE_WORKGROUP, true><<<blocks, threads>>>(tab, n_cells, per_thread, loc);
                    if (v == 2) k_scatter<99, false><<<blocks, threads>>>(tab, n_cells, per_thread, loc);
                    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
                }
            }
            printf("table %4u MB locality %3u: agent-scope %.1f Gop/s | wg-scope per-XCD replica %.1f Gop/s | plain 8-byte stores %.1f Gop/s\n",
                   mb, loc, n_ops / ms[0] * 1e-6, n_ops / ms[1] * 1e-6, n_ops / ms[2] * 1e-6);
        }
        hipFree(tab);
    }
    return 0;
}
