// LDS fragment-read microbenchmark: how many cycles does ONE wavefront pay per ds_read_b128, as a function of how many wavefronts
// of the CU read at the same time and how many reads it keeps in flight?  One workgroup per CU, WAVES wavefronts, every
// wavefront reads conflict-free 1 KB rows (lane * 16 within a 2 KB window, like an MFMA fragment of 32 rows x 64 bytes).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/lds_micro.hip -o tools/micro/lds_micro
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int INFLIGHT>
__global__ void lds_kernel(int iters, unsigned* sink, long long* cycles) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 1024 / 16; i += blockDim.x) ((uint4*)smem)[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    // fragment-like address: row = lane & 31 (64-byte rows), 16-byte slot (khalf ^ swizzle)
    const unsigned base = (wave * 4096u) % (64 * 1024 - 16 * 2048) + (lane & 31) * 64 + (((lane >> 5) ^ ((lane >> 2) & 3)) * 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint4 v[INFLIGHT];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) v[q] = *(const uint4*)(smem + base + ((it * INFLIGHT + q) & 15) * 2048);
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) { acc.x ^= v[q].x; acc.y += v[q].y; acc.z ^= v[q].z; acc.w += v[q].w; }
    }
    const long long t1 = clock64();
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678u) sink[0] = acc.x;
    if (lane == 0 && blockIdx.x == 0) cycles[wave] = t1 - t0;
}

template <int INFLIGHT>
void run(int waves, unsigned* sink, long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((lds_kernel<INFLIGHT>), dim3(256), dim3(waves * 64), 0, 0, 10, sink, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((lds_kernel<INFLIGHT>), dim3(256), dim3(waves * 64), 0, 0, iters, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double reads = (double)iters * INFLIGHT;
    printf("waves %2d, %2d reads in flight: %.1f clock64 ticks per read per wave (wave 0), kernel %.3f ms -> %.1f B/clk/CU at 2.4 GHz\n",
           waves, INFLIGHT, (double)h[0] / reads, ms, reads * waves * 1024.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
    unsigned* sink; long long* cyc;
    hipMalloc(&sink, 64); hipMalloc(&cyc, 16 * sizeof(long long));
    for (int waves : {1, 2, 4, 8, 12, 16}) {
        run<2>(waves, sink, cyc);
        run<4>(waves, sink, cyc);
        run<8>(waves, sink, cyc);
        run<12>(waves, sink, cyc);
    }
    return 0;
}
