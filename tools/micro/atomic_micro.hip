// fp32 atomic-add microbenchmark for the split reduction of the weight-gradient kernels: 504 workgroups each add a
// 128 x 128 fp32 tile (64 values per lane) into one of 36 tiles of a 2.36 MB matrix.
//   A  agent-scope atomics (what unsafeAtomicAdd emits: executed at the memory side, every XCD may touch every address)
//   B  workgroup-scope atomics (executed in the XCD's own L2) with every tile owned by ONE XCD: a workgroup reads HW_REG_XCC_ID and only
//      adds into tiles t with t % 8 == its XCC id.  Correct only because no other XCD touches those lines during the kernel; the
//      end-of-kernel release writes the L2 back.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_micro.hip -o tools/micro/atomic_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static __device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

template <int MODE>
__global__ __launch_bounds__(256) void atom_kernel(float* dst, int ntiles, int* hits, int* xcc_of_block) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcc = xcc_id();
    int tile;
    if (MODE == 0) tile = blockIdx.x % ntiles;
    else {
        const int per = (ntiles + 7) / 8;                 // tiles xcc, xcc + 8, ...
        tile = xcc + 8 * ((blockIdx.x / 8) % per);
        if (tile >= ntiles) tile = xcc;
    }
    if (tid == 0) { atomicAdd(hits + tile, 1); xcc_of_block[blockIdx.x] = xcc; }
    float* base = dst + (size_t)tile * 128 * 128;
    // the conv_wgrad_tr epilogue shape: wave (wc, wn) owns a 64 x 64 quadrant as 2 x 2 MFMA tiles; an instruction covers 2 rows x 32 floats
    const int wc = wave & 1, wn = wave >> 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wc * 64 + j * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                const int col = wn * 64 + i * 32 + (lane & 31);
                float* q = base + row * 128 + col;
                if (MODE == 0) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
}

int main() {
    const int ntiles = 36, nblk = 504;
    float* dst; int* hits; int* xob;
    hipMalloc(&dst, (size_t)ntiles * 128 * 128 * 4);
    hipMalloc(&hits, ntiles * 4);
    hipMalloc(&xob, nblk * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(dst, 0, (size_t)ntiles * 128 * 128 * 4);
        hipMemset(hits, 0, ntiles * 4);
        if (mode == 0) hipLaunchKernelGGL(atom_kernel<0>, dim3(nblk), dim3(256), 0, 0, dst, ntiles, hits, xob);
        else hipLaunchKernelGGL(atom_kernel<1>, dim3(nblk), dim3(256), 0, 0, dst, ntiles, hits, xob);
        hipDeviceSynchronize();
        std::vector<float> h((size_t)ntiles * 128 * 128);
        std::vector<int> hh(ntiles), hx(nblk);
        hipMemcpy(h.data(), dst, h.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hh.data(), hits, ntiles * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xob, nblk * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int t = 0; t < ntiles; ++t)
            for (int e = 0; e < 128 * 128; ++e) bad += h[(size_t)t * 16384 + e] != (float)hh[t];
        int rr = 0;
        for (int b = 0; b < nblk; ++b) rr += hx[b] == b % 8;
        int mn = 1 << 30, mx = 0;
        for (int t = 0; t < ntiles; ++t) { mn = hh[t] < mn ? hh[t] : mn; mx = hh[t] > mx ? hh[t] : mx; }
        const int iters = 20;
        hipEventRecord(e0);
        for (int it = 0; it < iters; ++it) {
            if (mode == 0) hipLaunchKernelGGL(atom_kernel<0>, dim3(nblk), dim3(256), 0, 0, dst, ntiles, hits, xob);
            else hipLaunchKernelGGL(atom_kernel<1>, dim3(nblk), dim3(256), 0, 0, dst, ntiles, hits, xob);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.1f us per launch of %d workgroups x 16384 atomics; wrong elements after one launch %ld; blocks with xcc == bid %% 8: %d / %d; "
               "workgroups per tile %d..%d\n", mode, mode ? "workgroup scope, tiles owned by one XCD" : "agent scope", ms * 1000 / iters, nblk, bad, rr, nblk, mn, mx);
    }
    return 0;
}
