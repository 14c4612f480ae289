// Infinity Cache (256 MiB, memory side) microbenchmark: does a consumer that walks its producer's output in the REVERSE order of the
// producer's writes get the tail of that tensor from the cache?  The step's layer1 / layer2 tensors (103 .. 411 MB) are written by one
// launch and read by the next; every kernel of the build walks its rows in ascending order, which is the worst case for an LRU-like
// cache smaller than the tensor (the oldest lines are evicted just before they are asked for).
//   producer: block-contiguous copy SRC -> A, ascending block order
//   consumer: (a) copy A -> B, (b) read-only sum of A; ascending or descending block order
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mall_micro.hip -o tools/micro/mall_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every block owns `per` consecutive v4u (a multiple of 256 * 4); NT: non-temporal loads / stores
template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(v4u* __restrict__ dst, const v4u* __restrict__ src, size_t per, int reverse) {
    const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const v4u* s = src + b * per;
    v4u* d = dst + b * per;
    for (size_t i = threadIdx.x; i < per; i += 256 * 4) {
        v4u v[4] = {};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < per) v[k] = NT ? __builtin_nontemporal_load(s + i + k * 256) : s[i + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k * 256 >= per) break;
            if (NT) __builtin_nontemporal_store(v[k], d + i + k * 256);
            else d[i + k * 256] = v[k];
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void sum_kernel(unsigned* __restrict__ out, const v4u* __restrict__ src, size_t per, int reverse) {
    const size_t b = reverse ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const v4u* s = src + b * per;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < per; i += 256 * 4) {
        v4u v[4] = {};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < per) v[k] = NT ? __builtin_nontemporal_load(s + i + k * 256) : s[i + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < per) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t MB = 1u << 20;
    const size_t sizes_mb[] = {96, 192, 392, 784};
    const size_t maxb = 784 * MB;
    v4u *src, *a, *b, *flush;
    unsigned* out;
    CK(hipMalloc(&src, maxb)); CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&flush, 1024 * MB)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, maxb)); CK(hipMemset(a, 2, maxb)); CK(hipMemset(b, 3, maxb)); CK(hipMemset(flush, 4, 1024 * MB));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("producer: copy SRC -> A ascending.  consumer times in us (GB/s of the bytes the consumer itself moves), median of 7\n");
    printf("%8s %6s %6s | %22s %22s | %22s %22s\n", "size MB", "nt", "blocks", "copy A->B ascending", "copy A->B descending", "sum(A) ascending", "sum(A) descending");
    for (size_t smb : sizes_mb) {
        const size_t bytes = smb * MB, n16 = bytes / 16;
        for (int nt = 0; nt < 2; ++nt) {
            for (int blocks : {2048, 16384}) {
                const size_t per = n16 / blocks;
                float res[4];
                for (int mode = 0; mode < 4; ++mode) {
                    float ts[7];
                    for (int rep = 0; rep < 7; ++rep) {
                        // push everything else out of the cache, then produce A ascending
                        hipLaunchKernelGGL(copy_kernel<false>, dim3(16384), dim3(256), 0, 0, flush, flush + 512 * MB / 16, (512 * MB / 16) / 16384, 0);
                        if (nt) hipLaunchKernelGGL(copy_kernel<true>, dim3(blocks), dim3(256), 0, 0, a, src, per, 0);
                        else hipLaunchKernelGGL(copy_kernel<false>, dim3(blocks), dim3(256), 0, 0, a, src, per, 0);
                        CK(hipEventRecord(e0, 0));
                        const int rev = mode & 1;
                        if (mode < 2) {
                            if (nt) hipLaunchKernelGGL(copy_kernel<true>, dim3(blocks), dim3(256), 0, 0, b, a, per, rev);
                            else hipLaunchKernelGGL(copy_kernel<false>, dim3(blocks), dim3(256), 0, 0, b, a, per, rev);
                        } else {
                            if (nt) hipLaunchKernelGGL(sum_kernel<true>, dim3(blocks), dim3(256), 0, 0, out, a, per, rev);
                            else hipLaunchKernelGGL(sum_kernel<false>, dim3(blocks), dim3(256), 0, 0, out, a, per, rev);
                        }
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ts[rep], e0, e1));
                    }
                    for (int i = 0; i < 7; ++i) for (int j = i + 1; j < 7; ++j) if (ts[j] < ts[i]) { float t = ts[i]; ts[i] = ts[j]; ts[j] = t; }
                    res[mode] = ts[3] * 1000.f;
                }
                printf("%8zu %6d %6d | %10.1f (%8.0f) %10.1f (%8.0f) | %10.1f (%8.0f) %10.1f (%8.0f)\n", smb, nt, blocks,
                       res[0], 2.0 * bytes / res[0] / 1e3, res[1], 2.0 * bytes / res[1] / 1e3, res[2], 1.0 * bytes / res[2] / 1e3, res[3], 1.0 * bytes / res[3] / 1e3);
            }
        }
    }
    // a three-kernel chain as the step has them: A -> B -> A -> B ..., all ascending against alternating directions
    for (size_t smb : {(size_t)192, (size_t)392}) {
        const size_t bytes = smb * MB, n16 = bytes / 16;
        const int blocks = 4096;
        const size_t per = n16 / blocks;
        for (int alt = 0; alt < 2; ++alt) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(copy_kernel<false>, dim3(16384), dim3(256), 0, 0, flush, flush + 512 * MB / 16, (512 * MB / 16) / 16384, 0);
                CK(hipEventRecord(e0, 0));
                for (int k = 0; k < 8; ++k) {
                    const int rev = alt ? (k & 1) : 0;
                    hipLaunchKernelGGL(copy_kernel<false>, dim3(blocks), dim3(256), 0, 0, (k & 1) ? a : b, (k & 1) ? b : a, per, rev);
                }
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (t < best) best = t;
            }
            printf("chain of 8 copies, %zu MB each, %s: %.1f us per copy (%.0f GB/s)\n", smb, alt ? "alternating directions" : "all ascending", best * 1000.f / 8,
                   2.0 * bytes / (best / 8 * 1e-3) / 1e9);
        }
    }
    return 0;
}
