// Which of two kernels on two streams gets the GPU's wave slots: the one enqueued first, or whichever holds them?
// A = long kernel whose waves use 128 VGPRs (fills every SIMD's register file, like the shading kernel);
// B = many small light blocks with a chain of dependent loads (like the geometry kernels).
// Prints B's duration alone, enqueued after A has started, and enqueued just before A.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef ITERS
#define ITERS 300
#endif

__global__ void __launch_bounds__(256, 4) kernel_a(float* out, int iters) {
    float v[96];
#pragma unroll
    for (int i = 0; i < 96; i++) v[i] = (float)(threadIdx.x + i);
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int i = 0; i < 96; i++) v[i] = fmaf(v[i], 1.0001f, v[(i + 1) % 96] * 0.5f);
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 96; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) kernel_b(const unsigned* __restrict__ chain, unsigned n, unsigned* out) {
    unsigned i = (blockIdx.x * 256u + threadIdx.x) % n;
    for (int k = 0; k < 20; k++) i = chain[i];
    out[blockIdx.x * 256 + threadIdx.x] = i;
}

int main() {
    const int blocks_a = 32768, blocks_b = 3072;
    const unsigned n = 1u << 20;
    float* oa; unsigned *chain, *ob;
    hipMalloc(&oa, (size_t)blocks_a * 256 * 4); hipMalloc(&chain, n * 4); hipMalloc(&ob, (size_t)blocks_b * 256 * 4);
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; i++) h[i] = (unsigned)(((unsigned long long)i * 2654435761ull + 12345ull) % n);
    hipMemcpy(chain, h.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t a0, a1, b0, b1; hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    auto run = [&](int mode) {   // 0: B alone, 1: A then (after it started) B, 2: B first then A, 3: A alone
        hipDeviceSynchronize();
        if (mode == 1 || mode == 3) { hipEventRecord(a0, s1); hipLaunchKernelGGL(kernel_a, dim3(blocks_a), dim3(256), 0, s1, oa, ITERS); hipEventRecord(a1, s1); }
        if (mode == 1) { hipStreamQuery(s1); for (volatile int w = 0; w < 2000000; w++) {} }
        if (mode != 3) { hipEventRecord(b0, s2); hipLaunchKernelGGL(kernel_b, dim3(blocks_b), dim3(256), 0, s2, chain, n, ob); hipEventRecord(b1, s2); }
        if (mode == 2) { hipEventRecord(a0, s1); hipLaunchKernelGGL(kernel_a, dim3(blocks_a), dim3(256), 0, s1, oa, ITERS); hipEventRecord(a1, s1); }
        hipDeviceSynchronize();
        float ta = 0, tb = 0, ab = 0;
        if (mode != 0) hipEventElapsedTime(&ta, a0, a1);
        if (mode != 3) hipEventElapsedTime(&tb, b0, b1);
        if (mode == 1 || mode == 2) hipEventElapsedTime(&ab, a0, b1);
        printf("mode %d: A %.3f ms, B %.3f ms, A start -> B end %.3f ms\n", mode, ta, tb, ab);
    };
    for (int rep = 0; rep < 2; rep++) for (int m : {3, 0, 1, 2}) run(m);
    return 0;
}
