// Linear-blend skinning of the vertex arena, once per frame before any raster view reads it.
//
// The reference has no skinning (its loader reads POSITION/NORMAL/TANGENT/TEXCOORD_0 only, Renderer/main.cpp:61-177,
// and animates node transforms alone, Renderer/Node.cpp:21-38); the vertex-stage conventions it does have are kept:
// normal, tangent and bitangent go through the upper 3x3 of the matrix like they go through `model` in
// Renderer/shaders/basic.vert:41-44, and are normalised there, not here.
//
// One lane per skinned vertex. Per vertex the kernel reads 96 bytes (rest position 16, rest attributes 48, joints 8,
// weights 16, destination + palette base 8) and writes 64 (position 16, attributes 48); the palette is small and
// stays in cache. The arithmetic order is fixed (and contraction is off) so that the CPU checker, which evaluates
// the same expression, produces the same bits:
//   S = ((w0*J[j0] + w1*J[j1]) + w2*J[j2]) + w3*J[j3]   per element of the three affine rows
//   p' = ((S_r0*x + S_r1*y) + S_r2*z) + S_r3,  d' = (S_r0*x + S_r1*y) + S_r2*z
#include "device_common.h"
#include "kernels.h"

namespace {

__global__ void __launch_bounds__(256) k_skin(const float4* __restrict__ rest_pos, const float4* __restrict__ rest_attr,
                                              const uint2* __restrict__ joints, const float4* __restrict__ weights,
                                              const uint2* __restrict__ dst_pal, const float4* __restrict__ palette,
                                              float4* __restrict__ pos, float4* __restrict__ attr, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint2 dp = dst_pal[i];
    const uint2 jj = joints[i];
    const float4 w4 = weights[i];
    const uint32_t j[4] = {jj.x & 0xFFFFu, jj.x >> 16, jj.y & 0xFFFFu, jj.y >> 16};
    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
    float S[3][4];  // S[row][col]
#pragma unroll
    for (int col = 0; col < 4; col++) {
        const float4 c0 = palette[(size_t)(dp.y + j[0]) * 4 + col], c1 = palette[(size_t)(dp.y + j[1]) * 4 + col];
        const float4 c2 = palette[(size_t)(dp.y + j[2]) * 4 + col], c3 = palette[(size_t)(dp.y + j[3]) * 4 + col];
        S[0][col] = ((w[0] * c0.x + w[1] * c1.x) + w[2] * c2.x) + w[3] * c3.x;
        S[1][col] = ((w[0] * c0.y + w[1] * c1.y) + w[2] * c2.y) + w[3] * c3.y;
        S[2][col] = ((w[0] * c0.z + w[1] * c1.z) + w[2] * c2.z) + w[3] * c3.z;
    }
    const float4 p = rest_pos[i];
    const float4 a0 = rest_attr[(size_t)i * 3], a1 = rest_attr[(size_t)i * 3 + 1], a2 = rest_attr[(size_t)i * 3 + 2];
    // DVertexAttr: nx ny nz tx | ty tz bx by | bz u v pad
    float o[3][3];  // [vector: normal, tangent, bitangent][component]
    const float vin[3][3] = {{a0.x, a0.y, a0.z}, {a0.w, a1.x, a1.y}, {a1.z, a1.w, a2.x}};
#pragma unroll
    for (int v = 0; v < 3; v++)
#pragma unroll
        for (int r = 0; r < 3; r++) o[v][r] = (S[r][0] * vin[v][0] + S[r][1] * vin[v][1]) + S[r][2] * vin[v][2];
    float4 q;
    q.x = ((S[0][0] * p.x + S[0][1] * p.y) + S[0][2] * p.z) + S[0][3];
    q.y = ((S[1][0] * p.x + S[1][1] * p.y) + S[1][2] * p.z) + S[1][3];
    q.z = ((S[2][0] * p.x + S[2][1] * p.y) + S[2][2] * p.z) + S[2][3];
    q.w = 1.0f;
    pos[dp.x] = q;
    attr[(size_t)dp.x * 3] = make_float4(o[0][0], o[0][1], o[0][2], o[1][0]);
    attr[(size_t)dp.x * 3 + 1] = make_float4(o[1][1], o[1][2], o[2][0], o[2][1]);
    attr[(size_t)dp.x * 3 + 2] = make_float4(o[2][2], a2.y, a2.z, 0.0f);
}

}  // namespace

void launch_skin(hipStream_t s, const void* rest_pos, const void* rest_attr, const void* joints, const void* weights, const void* dst_pal,
                 const float* palette, void* pos, void* attr, uint32_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_skin, dim3((n + 255) / 256), dim3(256), 0, s, (const float4*)rest_pos, (const float4*)rest_attr, (const uint2*)joints,
                       (const float4*)weights, (const uint2*)dst_pal, (const float4*)palette, (float4*)pos, (float4*)attr, n);
}
