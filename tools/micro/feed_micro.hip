// Operand-feed microbenchmark: how many bytes per clock per CU reach a workgroup from L2 when they come
//   (a) by LDS-DMA only (buffer_load ... lds), (b) by LDS-DMA plus ordinary global loads into registers, (c) registers only.
// Every workgroup re-reads a small L2-resident region, 4 workgroups of 256 threads per CU, like the 128x128 conv tile.
// Build: hipcc --offload-arch=gfx950 -O3 -I vince_amd/csrc tools/micro/feed_micro.hip -o tools/micro/feed_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
void vince_set_error(const char*, ...) {}

template <int DMA_PIECES, int REG_LOADS>
__global__ __launch_bounds__(256, 4) void feed_kernel(const void* src, uint32_t src_bytes, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rsrc = make_rsrc(src, src_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    uint32_t off = ((blockIdx.x * 7919u) % 64u) * 16384u + (uint32_t)tid * 16u;   // a 16 KB window inside a 1 MB region
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* gsrc = (const uint4*)src;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int pc = 0; pc < DMA_PIECES; ++pc)   // each piece: 64 lanes x 16 B = 1 KB per wave -> 4 KB per workgroup
            lds_dma16(__builtin_amdgcn_readfirstlane(smem_base + buf * 16384 + (pc * 4 + wave) * 1024), (off + pc * 4096u) % src_bytes, rsrc);
        uint4 r[REG_LOADS > 0 ? REG_LOADS : 1];
#pragma unroll
        for (int q = 0; q < REG_LOADS; ++q) r[q] = gsrc[((off + 65536u + q * 4096u) % src_bytes) / 16];
#pragma unroll
        for (int q = 0; q < REG_LOADS; ++q) { acc.x ^= r[q].x; acc.y += r[q].y; }
        wait_vmcnt<0>();
        __syncthreads();
        acc.z += *(const uint32_t*)(smem + buf * 16384 + tid * 4);
        off = (off + 16384u * 3u) % src_bytes;
    }
    if (acc.x + acc.y + acc.z == 0x12345678u) sink[0] = acc.x;
}

template <int D, int R>
double run(const void* src, uint32_t bytes, unsigned* sink, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed_kernel<D, R>), dim3(1024), dim3(256), 0, 0, src, bytes, 50, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed_kernel<D, R>), dim3(1024), dim3(256), 0, 0, src, bytes, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = 4.0 * iters * (D * 4096.0 + R * 4096.0);   // 4 workgroups per CU
    const double clk = ms * 1e-3 * 2.4e9;
    printf("DMA %2d KB + registers %2d KB per workgroup-iteration: %.2f ms, %.1f B/clk/CU at 2.4 GHz (%.1f TB/s chip)\n",
           D * 4, R * 4, ms, bytes_per_cu / clk, bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
    return ms;
}


// Gather pattern of the implicit-GEMM operand fetch: every wave instruction brings ROWB-byte pieces of 1024/ROWB different
// rows (row stride `stride` bytes, like Ci * 2 of an NHWC activation), K steps walk along the row.  ROWB = 64 -> half cache
// lines (the other half comes with the next K step); ROWB = 128 -> whole lines.  Same bytes per iteration either way.
template <int ROWB, int PIECES>
__global__ __launch_bounds__(256, 4) void gather_kernel(const void* src, uint32_t src_bytes, uint32_t stride, int ksteps, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PIECES * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rsrc = make_rsrc(src, src_bytes);
    const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    constexpr int LPR = ROWB / 16, RPW = 64 / LPR;            // lanes per row, rows per wave instruction
    const uint32_t rows_per_iter = PIECES * 4 * RPW;
    uint32_t acc = 0;
    uint32_t row0 = (blockIdx.x * 7919u) % 4096u;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        const int ks = it % ksteps;
        if (ks == 0) row0 = (row0 + rows_per_iter * 13u);
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
            const uint32_t row = row0 + (pc * 4 + wave) * RPW + lane / LPR;
            const uint32_t off = (row * stride + ks * ROWB + (lane % LPR) * 16) % src_bytes;
            lds_dma16(__builtin_amdgcn_readfirstlane(smem_base + (buf * PIECES * 4 + pc * 4 + wave) * 1024), off, rsrc);
        }
        wait_vmcnt<0>();
        __syncthreads();
        acc += *(const uint32_t*)(smem + buf * PIECES * 4096 + tid * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int ROWB, int PIECES>
void run_gather(const void* src, uint32_t bytes, unsigned* sink, uint32_t stride, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int ksteps = stride / ROWB;
    hipLaunchKernelGGL((gather_kernel<ROWB, PIECES>), dim3(1024), dim3(256), 0, 0, src, bytes, stride, ksteps, 50, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather_kernel<ROWB, PIECES>), dim3(1024), dim3(256), 0, 0, src, bytes, stride, ksteps, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = 4.0 * iters * PIECES * 4096.0;
    const double clk = ms * 1e-3 * 2.4e9;
    printf("gather: %3d-byte row pieces, row stride %5u B, %d KB per workgroup-iteration: %.2f ms, %.1f B/clk/CU (%.1f TB/s chip)\n",
           ROWB, stride, PIECES * 4, ms, bytes_per_cu / clk, bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const uint32_t bytes = 1u << 20;
    void* src; unsigned* sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes);
    const int iters = 2000;
    run<4, 0>(src, bytes, sink, iters);
    run<4, 2>(src, bytes, sink, iters);
    run<6, 0>(src, bytes, sink, iters);
    run<2, 2>(src, bytes, sink, iters);
    run<0, 4>(src, bytes, sink, iters);
    run<0, 6>(src, bytes, sink, iters);
    run<4, 4>(src, bytes, sink, iters);
    run<8, 0>(src, bytes, sink, iters);
    {
        const uint32_t big = 256u << 20;   // 256 MB: rows come from L2 / Infinity Cache / HBM like an activation tensor
        void* act; hipMalloc(&act, big); hipMemset(act, 1, big);
        for (uint32_t stride : {512u, 2048u, 4608u}) {
            run_gather<64, 4>(act, big, sink, stride, iters);
            run_gather<128, 4>(act, big, sink, stride, iters);
            run_gather<64, 6>(act, big, sink, stride, iters);
            run_gather<128, 6>(act, big, sink, stride, iters);
        }
        // L2-resident rows (weights-like): 1 MB region
        run_gather<64, 4>(src, bytes, sink, 512, iters);
        run_gather<128, 4>(src, bytes, sink, 512, iters);
    }
    return 0;
}
