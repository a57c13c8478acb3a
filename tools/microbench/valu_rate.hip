// Microbenchmark: how many cycles does a wave64 VALU instruction cost a SIMD of gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Chains of independent f32 fmas (16 accumulators per lane), of v_pk_fma_f32 (8 packed accumulators) and of dependent
// fmas (1 accumulator), with 1, 2, 4 and 8 waves per SIMD.  Reports SIMD cycles per wave instruction (at the clock
// hipDeviceProp reports).  Used to read the VALU counters of the instruction-bound kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>   // 0: 16 independent fma chains (the compiler packs them), 1: 8 packed chains, 2: one dependent chain, 3: 16 unpacked
__global__ __launch_bounds__(256) void k(float *out, uint32_t iters, float a, float b) {
    float x[16];
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = f2{(float)(threadIdx.x + i), (float)i};
    for (uint32_t it = 0; it < iters; it++) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) p[i] = __builtin_elementwise_fma(p[i], f2{a, a}, f2{b, b});
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[0] = __builtin_fmaf(x[0], a, b);
        } else {        // 16 independent v_fma_f32 the compiler cannot pack
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const double clk = prop.clockRate * 1e3;      // Hz
    const int cus = prop.multiProcessorCount;
    float *out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t iters = 20000;
    printf("%d CUs at %.0f MHz\n", cus, clk / 1e6);
    for (int kind = 0; kind < 4; kind++)
        for (int waves_per_simd : {1, 2, 4, 8}) {
            const int blocks = cus * waves_per_simd;        // a block = 4 waves = one per SIMD
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)waves_per_simd * iters * (kind == 0 ? 8.0 : 16.0);
            printf("%-26s %d waves/SIMD: %8.3f ms -> %.2f SIMD cycles per wave instruction\n",
                   kind == 0 ? "16 fma as 8 v_pk_fma_f32" : (kind == 1 ? "16 v_pk_fma_f32 (8 chains)" : (kind == 2 ? "16 dependent v_fma_f32" : "16 independent v_fma_f32")),
                   waves_per_simd, best, best * 1e-3 * clk / instr_per_simd);
        }
    return 0;
}
