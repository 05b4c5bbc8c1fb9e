// Issue rate of plain and packed f32 VALU instructions on one SIMD with 1, 2, 4 waves resident:
// every wave runs a loop of 16 independent fma chains; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float x[16];
    v2f y[8];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (float)threadIdx.x + i;
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = v2f{(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(y[i], v2f{a, a}, v2f{b, b});
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_amdgcn_rcpf(x[i]);
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = (float)(((int)x[i] * 3 + 7) & 0xFFFF);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int per_iter) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 16 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks_per_cu : {1, 2, 4, 8}) {  // a 256-thread block = one wave per SIMD
        const int grid = 256 * blocks_per_cu;
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * per_iter * blocks_per_cu;  // wave-instructions issued on one SIMD
        printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, blocks_per_cu, ms,
               ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    }
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_rcp_f32", 16);
    run<3>("int mul/add/cvt", 16 * 5);
    return 0;
}
