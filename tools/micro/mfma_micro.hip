// Sustained rate of the 16-bit matrix pipe on gfx950, nothing but MFMAs: how far below the nominal 2.5 PFLOP/s (256 CUs x 4 SIMDs x
// 1024 FLOP/clk x 2.4 GHz) a kernel lands that does NOTHING else -- the ceiling DESIGN.md prices the MFMA-bound convolutions against
// beside the nominal peak.  Every wavefront runs `iters` x 16 independent-accumulator v_mfma_f32_32x32x16 (bf16 or f16, 4 accumulators
// round-robin: no MFMA waits for its predecessor); grid = CUs x workgroups per CU, 256 or 512 threads (1 or 2 wavefronts per SIMD).
// Reports TFLOP/s and the clock the matrix pipe must have run at if it never idled (32 cycles per MFMA).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_micro.hip -o tools/micro/mfma_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16v8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <bool HALF, int NACC>
__global__ void mfma_kernel(float* out, int iters) {
    f32x16_t acc[4];
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    bf16v8_t xb, yb;
    f16x8_t xh, yh;
    for (int e = 0; e < 8; ++e) {
        xb[e] = (__bf16)(float)(threadIdx.x + e);
        yb[e] = (__bf16)(float)(blockIdx.x + e);
        xh[e] = (_Float16)(float)((threadIdx.x + e) & 15);
        yh[e] = (_Float16)(float)((blockIdx.x + e) & 15);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (HALF) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, acc[u % NACC], 0, 0, 0);
            else acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[u % NACC], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int e = 0; e < 16; ++e) s += acc[a][e];
    if (s == 1.2345f) out[0] = s;
}

int main(int argc, char** argv) {
    float* out;
    hipMalloc(&out, 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%d CUs, nominal clock %.2f GHz; 32 x 32 x 16 MFMAs only\n", cus, prop.clockRate / 1e6);
    printf("%-6s %8s %8s %10s %10s %10s %12s\n", "type", "threads", "CUs used", "iters", "ms", "TFLOP/s", "implied GHz");
    for (int half = 0; half < 2; ++half)
        for (int threads : {256, 512})
            for (int used : {cus, (cus * 196) / 256})
                for (int iters : {400, 4000, 40000}) {
                    for (int rep = 0; rep < 2; ++rep) {   // first = warm-up
                        hipEventRecord(e0, 0);
                        if (half) hipLaunchKernelGGL((mfma_kernel<true, 4>), dim3(used), dim3(threads), 0, 0, out, iters);
                        else hipLaunchKernelGGL((mfma_kernel<false, 4>), dim3(used), dim3(threads), 0, 0, out, iters);
                        hipEventRecord(e1, 0);
                        hipEventSynchronize(e1);
                    }
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double mfmas_per_simd = (double)iters * 16 * (threads / 256);
                    const double flops = mfmas_per_simd * 4 * used * 32768.0;
                    printf("%-6s %8d %8d %10d %10.3f %10.1f %12.2f\n", half ? "f16" : "bf16", threads, used, iters, ms, flops / ms / 1e9,
                           mfmas_per_simd * 32 / (ms * 1e6));
                }
    // dependent accumulation: every MFMA adds into one of NACC accumulators (round-robin), f16, all CUs, 4000 iterations
    printf("accumulators in rotation (f16, %d CUs, 4000 iterations): distance between an MFMA and the one it depends on\n", cus);
    for (int threads : {256, 512})
        for (int nacc : {1, 2, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (nacc == 1) hipLaunchKernelGGL((mfma_kernel<true, 1>), dim3(cus), dim3(threads), 0, 0, out, 4000);
                else if (nacc == 2) hipLaunchKernelGGL((mfma_kernel<true, 2>), dim3(cus), dim3(threads), 0, 0, out, 4000);
                else hipLaunchKernelGGL((mfma_kernel<true, 4>), dim3(cus), dim3(threads), 0, 0, out, 4000);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double mfmas_per_simd = 4000.0 * 16 * (threads / 256);
            printf("threads %d  accumulators %d: %8.3f ms  %8.1f TFLOP/s  %.1f cycles per MFMA at 2.4 GHz\n", threads, nacc, ms,
                   mfmas_per_simd * 4 * cus * 32768.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
        }
    return 0;
}
