// What rocprofv3's FETCH_SIZE counts for the access shapes of the shading kernel: each kernel below reads a 2 GiB
// buffer (eight times the 256 MiB last-level cache) touching every 128-byte line once, in one shape:
//   stream16 / stream8 / stream4   neighbouring lanes read neighbouring 16 / 8 / 4 bytes (fragment records are 8)
//   gather8 / gather16             every lane reads 8 / 16 bytes of a line of its own (texel pairs, shadow-map pairs,
//                                  cube and table footprints), lines in a scrambled order
//   gather8_halves                 as gather8, but both 64-byte halves of every line, far apart in time
// Build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/fetch_calib tools/fetch_calib.hip
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- gpurun_out/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr size_t BYTES = size_t(2) << 30;
constexpr uint32_t LINES = uint32_t(BYTES / 128);  // a power of two: an odd multiplier permutes the lines

template <typename T>
__global__ void stream(const T* __restrict__ p, size_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = p[i];
        const uint32_t* w = (const uint32_t*)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345u) *sink = acc;  // keeps the loads
}
template <typename T, int HALVES>
__global__ void gather(const uint8_t* __restrict__ p, uint32_t* sink) {
    uint32_t acc = 0;
    const uint32_t total = LINES * (uint32_t)HALVES;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t line = (i * 0x9E3779B1u) & (LINES - 1u);
        const uint32_t half = HALVES == 2 ? i / LINES : 0u;  // the second halves come after all the first ones
        const T v = *(const T*)(p + (size_t)line * 128u + half * 64u + 16u);
        const uint32_t* w = (const uint32_t*)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345u) *sink = acc;
}

int main() {
    uint8_t* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, BYTES));
    CHECK(hipDeviceSynchronize());
    const dim3 grid(256 * 8), block(256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms[6];
    for (int rep = 0; rep < 2; rep++) {  // the second round is the one to read (the first pays for page faults and clocks)
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL(stream<uint4>, grid, block, 0, 0, (const uint4*)buf, BYTES / 16, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[0], a, b));
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL(stream<uint2>, grid, block, 0, 0, (const uint2*)buf, BYTES / 8, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[1], a, b));
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL(stream<uint32_t>, grid, block, 0, 0, (const uint32_t*)buf, BYTES / 4, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[2], a, b));
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL((gather<uint2, 1>), grid, block, 0, 0, buf, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[3], a, b));
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL((gather<uint4, 1>), grid, block, 0, 0, buf, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[4], a, b));
        CHECK(hipEventRecord(a)); hipLaunchKernelGGL((gather<uint2, 2>), grid, block, 0, 0, buf, sink); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms[5], a, b));
    }
    const char* names[6] = {"stream16", "stream8", "stream4", "gather8", "gather16", "gather8_halves"};
    for (int k = 0; k < 6; k++)
        printf("%-15s %8.3f ms  %7.1f M lines/s  (%u lines of 128 B, buffer %zu bytes)\n", names[k], ms[k], LINES * (k == 5 ? 1.0 : 1.0) / ms[k] * 1e-3, LINES, BYTES);
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
