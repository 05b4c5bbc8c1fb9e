// Post-process tail of the frame: colour -> bloom mip 0, five 13-tap
// down-samples, five 3x3 tent up-samples, filmic tonemap + log-luminance,
// centre-crop average luminance.
//
// Reference: Renderer/VulkanRendererBloom.cpp:62-100 (schedule),
// Renderer/shaders/bloomDownsample.comp, bloomUpsample.comp, tonemap.comp,
// averageLuminance.comp; Renderer/VulkanRenderer.cpp:1156-1191.
//
// All of these are HBM-bound: every pixel is touched once per pass. What is
// done about it:
//  * bloom mip 0 (the colour image converted to B10G11R11) is never read back
//    from memory: the first down-sample and the last up-sample take the colour
//    image itself and apply the storage truncation in registers;
//  * the last up-sample is fused into the tonemap kernel (tonemap needs only
//    mip 0's final value), which still stores mip 0 for readers of the chain;
//  * the up-sample reads its 3x3 source taps at (x/2, y/2): the four pixels of
//    a 2x2 quad share them, so one thread handles a quad and decodes the nine
//    packed taps once instead of four times;
//  * threads map to pixels so that a wave touches whole rows (coalesced
//    segments); neighbour re-reads are served by L1/L2.
// Arithmetic order (fma chains in tap order, packed-float truncation after
// every pass) is exactly the pass-by-pass order of the reference chain, so
// every stored mip matches the unfused chain bit for bit.
#include "device_common.h"
#include "kernels.h"

namespace {

RDR_DEV int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// value of a colour-image texel as bloom mip 0 stores it (B10G11R11, truncated)
RDR_DEV void load_mip0_from_hdr(const uint2* __restrict__ hdr, size_t i, float& r, float& g, float& b) {
    float a;
    unpack_half4(hdr[i], r, g, b, a);
    unpack_r11g11b10(pack_r11g11b10(r, g, b), r, g, b);
}

// bloomDownsample.comp:14-33. The LINEAR unnormalized sampler is hit exactly
// on texel centres, so each tap is one texel; taps in shader order.
// FROM_HDR: the source level is mip 0, read from the colour image.
template <bool FROM_HDR>
__global__ void __launch_bounds__(256) k_bloom_down(const void* __restrict__ src_v, int sw, int sh, uint32_t* __restrict__ dst,
                                                    int dw, int dh, int dst_pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
    const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
    const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 13; t++) {
        const int sx = clampi(2 * x + dxs[t], 0, sw - 1), sy = clampi(2 * y + dys[t], 0, sh - 1);
        float r, g, b;
        if (FROM_HDR) load_mip0_from_hdr((const uint2*)src_v, (size_t)sy * sw + sx, r, g, b);
        else unpack_r11g11b10(((const uint32_t*)src_v)[(size_t)sy * sw + sx], r, g, b);
        acc0 = fmaf(ws[t], r, acc0); acc1 = fmaf(ws[t], g, acc1); acc2 = fmaf(ws[t], b, acc2);
    }
    dst[(size_t)y * dst_pitch + x] = pack_r11g11b10(acc0, acc1, acc2);
}

// the nine taps of bloomUpsample.comp:17-25 around source texel (ux, uy), decoded once
struct Tent { float r[9], g[9], b[9]; };
RDR_DEV void load_tent(const uint32_t* __restrict__ src, int sw, int sh, int ux, int uy, Tent& t) {
    const int dxs[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    const int dys[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int sx = clampi(ux + dxs[k], 0, sw - 1), sy = clampi(uy + dys[k], 0, sh - 1);
        unpack_r11g11b10(src[(size_t)sy * sw + sx], t.r[k], t.g[k], t.b[k]);
    }
}
// dst += tent3x3(src): the read-modify-write of one destination texel, fma chain in tap order
RDR_DEV uint32_t tent_accumulate(const Tent& t, float acc0, float acc1, float acc2) {
    const float ws[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
#pragma unroll
    for (int k = 0; k < 9; k++) { acc0 = fmaf(ws[k], t.r[k], acc0); acc1 = fmaf(ws[k], t.g[k], acc1); acc2 = fmaf(ws[k], t.b[k], acc2); }
    return pack_r11g11b10(acc0, acc1, acc2);
}

// bloomUpsample.comp:15-29, one thread per 2x2 quad of the destination level
__global__ void __launch_bounds__(256) k_bloom_up(const uint32_t* __restrict__ src, int sw, int sh, uint32_t* __restrict__ dst,
                                                  int dw, int dh, int dst_pitch) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= dw || 2 * qy >= dh) return;
    Tent t;
    load_tent(src, sw, sh, qx, qy, t);
#pragma unroll
    for (int oy = 0; oy < 2; oy++)
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
            const int x = 2 * qx + ox, y = 2 * qy + oy;
            if (x >= dw || y >= dh) continue;
            float a0, a1, a2;
            unpack_r11g11b10(dst[(size_t)y * dst_pitch + x], a0, a1, a2);
            dst[(size_t)y * dst_pitch + x] = tent_accumulate(t, a0, a1, a2);
        }
}

// ---- the same passes with their source tiles staged in LDS -------------------------------------------------------
//
// k_bloom_down_lds: a block produces DOWN_TX x DOWN_TY destination texels (two per thread) and first copies the
// (2 DOWN_TX + 4) x (2 DOWN_TY + 4) source texels its 13-tap footprints cover into LDS. A staged texel is three
// binary16 values (red and green in one dword, blue in the low half of a second): B10G11R11 is binary16 without the sign
// and the low mantissa bits, so the truncation the storage format applies is a mask on the half bits of the colour image
// (done once per texel instead of once per tap), a packed mip texel expands with three shifts, and a tap is one 8-byte
// LDS read plus three mixed-precision fmas that convert the half operand on the fly. Even and odd source columns are
// kept in separate halves of a row, so the lanes of a wave (destination x = lane) read consecutive LDS words. Edge
// clamping happens while staging. Values and the order of the fma chain are those of k_bloom_down: identical results.
#define DOWN_TX 32
#define DOWN_TY 16
#define DOWN_SW (2 * DOWN_TX + 4)
#define DOWN_SH (2 * DOWN_TY + 4)
#define DOWN_HW (DOWN_SW / 2)  // texels per column parity in a row

typedef _Float16 rdr_f16x2 __attribute__((ext_vector_type(2)));
RDR_DEV rdr_f16x2 as_f16x2(uint32_t u) { rdr_f16x2 h; __builtin_memcpy(&h, &u, 4); return h; }

// colour-image texel (four binary16) -> staged texel: what unpack(pack_r11g11b10()) of it would give, as half bits
RDR_DEV uint2 stage_from_hdr(uint2 t) {
    // any exponent field all ones (inf / NaN): the exact, slower route
    const uint32_t e0 = t.x & 0x7C007C00u, e1 = t.y & 0x7C00u;
    if (__builtin_expect((((e0 + 0x04000400u) & 0x80008000u) | ((e1 + 0x0400u) & 0x8000u)) != 0u, 0)) {
        const uint32_t r = ufloat_from_half_bits<6>(t.x & 0xFFFFu) << 4, g = ufloat_from_half_bits<6>(t.x >> 16) << 4;
        const uint32_t b = ufloat_from_half_bits<5>(t.y & 0xFFFFu) << 5;
        return make_uint2(r | (g << 16), b);
    }
    // finite: negative -> 0 (as 16-bit integers the negative halves are < 0), low mantissa bits dropped
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    s16x2 v0, v1;
    __builtin_memcpy(&v0, &t.x, 4); __builtin_memcpy(&v1, &t.y, 4);
    const s16x2 zero = {0, 0};
    v0 = __builtin_elementwise_max(v0, zero); v1 = __builtin_elementwise_max(v1, zero);
    uint32_t rg, b;
    __builtin_memcpy(&rg, &v0, 4); __builtin_memcpy(&b, &v1, 4);
    return make_uint2(rg & 0xFFF0FFF0u, b & 0x0000FFE0u);
}
RDR_DEV uint2 stage_from_mip(uint32_t p) {
    return make_uint2(((p & 0x7FFu) << 4) | ((p << 9) & 0x7FF00000u), (p >> 22) << 5);
}
// LDS word index of source column c (0 .. DOWN_SW-1) in row r
RDR_DEV int down_slot(int r, int c) { return r * DOWN_SW + (c & 1) * DOWN_HW + (c >> 1); }

// 13-tap kernel (bloomDownsample.comp:14-33, taps in shader order) for destination texel (tx, ty) of the tile
RDR_DEV uint32_t down13_staged(const uint2* tile, int tx, int ty) {
    const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
    const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
    const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
    const uint2* base = tile + (2 * ty + 2) * DOWN_SW + tx + 1;  // column 2 tx + 2: even, index tx + 1 of its half row
    float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 13; t++) {
        // column 2 tx + 2 + dx: parity of dx, index tx + 1 + floor(dx / 2)
        const int off = dys[t] * DOWN_SW + (dxs[t] & 1) * DOWN_HW + (dxs[t] >> 1);
        const uint2 v = base[off];
        const rdr_f16x2 rg = as_f16x2(v.x), bb = as_f16x2(v.y);
        acc0 = fmaf(ws[t], (float)rg.x, acc0); acc1 = fmaf(ws[t], (float)rg.y, acc1); acc2 = fmaf(ws[t], (float)bb.x, acc2);
    }
    return pack_r11g11b10(acc0, acc1, acc2);
}

template <bool FROM_HDR>
__global__ void __launch_bounds__(256) k_bloom_down_lds(const void* __restrict__ src_v, int sw, int sh, uint32_t* __restrict__ dst,
                                                        int dw, int dh, int dst_pitch) {
    __shared__ uint2 tile[DOWN_SH * DOWN_SW];
    const int x0 = blockIdx.x * DOWN_TX, y0 = blockIdx.y * DOWN_TY;
    const int sx0 = 2 * x0 - 2, sy0 = 2 * y0 - 2;
    // the whole source tile inside the image and rows that keep pairs of texels aligned: two texels per load, no clamps
    const bool interior = sx0 >= 0 && sy0 >= 0 && sx0 + DOWN_SW <= sw && sy0 + DOWN_SH <= sh && (sw & 1) == 0;
    if (interior) {
        for (int i = threadIdx.x; i < DOWN_SH * DOWN_HW; i += 256) {
            const int ly = i / DOWN_HW, lp = i - ly * DOWN_HW;  // pair lp = columns 2 lp, 2 lp + 1
            const size_t at = (size_t)(sy0 + ly) * sw + sx0 + 2 * lp;
            uint2 e, o;
            if (FROM_HDR) {
                const uint4 v = *(const uint4*)((const uint2*)src_v + at);
                e = stage_from_hdr(make_uint2(v.x, v.y)); o = stage_from_hdr(make_uint2(v.z, v.w));
            } else {
                const uint2 v = *(const uint2*)((const uint32_t*)src_v + at);
                e = stage_from_mip(v.x); o = stage_from_mip(v.y);
            }
            tile[ly * DOWN_SW + lp] = e;
            tile[ly * DOWN_SW + DOWN_HW + lp] = o;
        }
    } else {
        for (int i = threadIdx.x; i < DOWN_SH * DOWN_SW; i += 256) {
            const int ly = i / DOWN_SW, lx = i - ly * DOWN_SW;
            const int sx = clampi(sx0 + lx, 0, sw - 1), sy = clampi(sy0 + ly, 0, sh - 1);
            tile[down_slot(ly, lx)] = FROM_HDR ? stage_from_hdr(((const uint2*)src_v)[(size_t)sy * sw + sx])
                                               : stage_from_mip(((const uint32_t*)src_v)[(size_t)sy * sw + sx]);
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & (DOWN_TX - 1), ty = threadIdx.x / DOWN_TX;
    const int x = x0 + tx;
    if (x >= dw) return;
#pragma unroll
    for (int k = 0; k < DOWN_TY / 8; k++) {
        const int y = y0 + ty + 8 * k;
        if (y < dh) dst[(size_t)y * dst_pitch + x] = down13_staged(tile, tx, ty + 8 * k);
    }
}

// k_bloom_down_tail: the three small down-sample passes (mip 2 -> 3 -> 4 -> 5) in one launch. A block owns a 4x4 tile of
// mip 5 and with it the 8x8 tile of mip 4 and the 16x16 tile of mip 3 above it. It stages the 60x60 texels of mip 2
// that those depend on, computes the 28x28 region of mip 3 and the 12x12 region of mip 4 it needs (their borders are
// computed again by the neighbouring blocks: same inputs, same operations, same bits), stores the tiles it owns and
// its mip 5 texels. Three launches of a few microseconds each, mostly launch latency, become one.
#define TAIL_T5 4
#define TAIL_R4 (2 * TAIL_T5 + 4)
#define TAIL_R3 (2 * TAIL_R4 + 4)
#define TAIL_R2 (2 * TAIL_R3 + 4)
struct BloomLevels { uint32_t* mip[6]; int w[6], h[6]; uint32_t *fin3, *fin4; };  // pointers, sizes (= row pitches) of the six mips; see k_bloom_up_tail
__global__ void __launch_bounds__(256) k_bloom_down_tail(BloomLevels L) {
    __shared__ uint32_t l2[TAIL_R2 * TAIL_R2], l3[TAIL_R3 * TAIL_R3], l4[TAIL_R4 * TAIL_R4];
    const int o5x = blockIdx.x * TAIL_T5, o5y = blockIdx.y * TAIL_T5;
    const int o4x = 2 * o5x - 2, o4y = 2 * o5y - 2, o3x = 2 * o4x - 2, o3y = 2 * o4y - 2, o2x = 2 * o3x - 2, o2y = 2 * o3y - 2;
    for (int i = threadIdx.x; i < TAIL_R2 * TAIL_R2; i += 256) {
        const int ly = i / TAIL_R2, lx = i - ly * TAIL_R2;
        l2[i] = L.mip[2][(size_t)clampi(o2y + ly, 0, L.h[2] - 1) * L.w[2] + clampi(o2x + lx, 0, L.w[2] - 1)];
    }
    __syncthreads();
    // mip 3 region: texel c = o3 + l reads mip 2 at clamp(2c + d), which sits at LDS index clamp(2c + d) - o2
    for (int i = threadIdx.x; i < TAIL_R3 * TAIL_R3; i += 256) {
        const int ly = i / TAIL_R3, lx = i - ly * TAIL_R3, cx = o3x + lx, cy = o3y + ly;
        uint32_t v = 0u;
        if (cx >= 0 && cy >= 0 && cx < L.w[3] && cy < L.h[3]) {
            const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
            const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
            const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int sx = clampi(2 * cx + dxs[t], 0, L.w[2] - 1) - o2x, sy = clampi(2 * cy + dys[t], 0, L.h[2] - 1) - o2y;
                float r, g, b;
                unpack_r11g11b10(l2[sy * TAIL_R2 + sx], r, g, b);
                a0 = fmaf(ws[t], r, a0); a1 = fmaf(ws[t], g, a1); a2 = fmaf(ws[t], b, a2);
            }
            v = pack_r11g11b10(a0, a1, a2);
            // the tile this block owns: mip 3 texels [4 o5, 4 o5 + 16)
            if (cx >= 4 * o5x && cx < 4 * o5x + 4 * TAIL_T5 && cy >= 4 * o5y && cy < 4 * o5y + 4 * TAIL_T5) L.mip[3][(size_t)cy * L.w[3] + cx] = v;
        }
        l3[i] = v;
    }
    __syncthreads();
    if (threadIdx.x < TAIL_R4 * TAIL_R4) {
        const int i = threadIdx.x, ly = i / TAIL_R4, lx = i - ly * TAIL_R4, cx = o4x + lx, cy = o4y + ly;
        uint32_t v = 0u;
        if (cx >= 0 && cy >= 0 && cx < L.w[4] && cy < L.h[4]) {
            const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
            const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
            const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int sx = clampi(2 * cx + dxs[t], 0, L.w[3] - 1) - o3x, sy = clampi(2 * cy + dys[t], 0, L.h[3] - 1) - o3y;
                float r, g, b;
                unpack_r11g11b10(l3[sy * TAIL_R3 + sx], r, g, b);
                a0 = fmaf(ws[t], r, a0); a1 = fmaf(ws[t], g, a1); a2 = fmaf(ws[t], b, a2);
            }
            v = pack_r11g11b10(a0, a1, a2);
            if (cx >= 2 * o5x && cx < 2 * o5x + 2 * TAIL_T5 && cy >= 2 * o5y && cy < 2 * o5y + 2 * TAIL_T5) L.mip[4][(size_t)cy * L.w[4] + cx] = v;
        }
        l4[i] = v;
    }
    __syncthreads();
    if (threadIdx.x < TAIL_T5 * TAIL_T5) {
        const int cx = o5x + (int)(threadIdx.x % TAIL_T5), cy = o5y + (int)(threadIdx.x / TAIL_T5);
        if (cx < L.w[5] && cy < L.h[5]) {
            const int dxs[13] = {+1, +1, -1, -1, -2, 0, +2, +2, +2, 0, -2, -2, 0};
            const int dys[13] = {+1, -1, +1, -1, -2, -2, -2, 0, +2, +2, +2, 0, 0};
            const float ws[13] = {0.125f, 0.125f, 0.125f, 0.125f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.03125f, 0.0625f, 0.125f};
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int t = 0; t < 13; t++) {
                const int sx = clampi(2 * cx + dxs[t], 0, L.w[4] - 1) - o4x, sy = clampi(2 * cy + dys[t], 0, L.h[4] - 1) - o4y;
                float r, g, b;
                unpack_r11g11b10(l4[sy * TAIL_R4 + sx], r, g, b);
                a0 = fmaf(ws[t], r, a0); a1 = fmaf(ws[t], g, a1); a2 = fmaf(ws[t], b, a2);
            }
            L.mip[5][(size_t)cy * L.w[5] + cx] = pack_r11g11b10(a0, a1, a2);
        }
    }
}

// 3x3 tent of bloomUpsample.comp around source texel (ux, uy) of a packed LDS region with origin (ox, oy), added onto
// the destination value `d` (read-modify-write of the shader), fma chain in tap order; source coordinates clamp to the
// source image.
RDR_DEV uint32_t tent_lds(const uint32_t* reg, int pitch, int ox, int oy, int sw, int sh, int ux, int uy, uint32_t d) {
    const int dxs[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    const int dys[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
    const float ws[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
    float a0, a1, a2;
    unpack_r11g11b10(d, a0, a1, a2);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int sx = clampi(ux + dxs[k], 0, sw - 1) - ox, sy = clampi(uy + dys[k], 0, sh - 1) - oy;
        float r, g, b;
        unpack_r11g11b10(reg[sy * pitch + sx], r, g, b);
        a0 = fmaf(ws[k], r, a0); a1 = fmaf(ws[k], g, a1); a2 = fmaf(ws[k], b, a2);
    }
    return pack_r11g11b10(a0, a1, a2);
}

// k_bloom_up_tail: the three small up-sample passes (mip 5 -> 4 -> 3 -> 2) in one launch. A block owns a 32x32 tile of
// mip 2 and the 16x16 / 8x8 tiles of mips 3 / 4 under it. It stages the 8x8 texels of mip 5, the 12x12 of mip 4 and
// the 18x18 of mip 3 its tile depends on, accumulates level by level in LDS (mip 4 += tent(mip 5), then mip 3 +=
// tent(mip 4), each truncated to the packed format as the separate passes store it), writes the tiles it owns and
// finally its mip 2 texels. Region borders are recomputed by the neighbours with identical results. The accumulated
// mips 3 and 4 go to arrays of their own (fin3, fin4): other blocks are still staging the down-sampled values of those
// texels from mip[3] / mip[4] while this one finishes. Mip 2 is updated in place: only its owner touches a texel.
#define UPT_T2 32
#define UPT_R3 (UPT_T2 / 2 + 2)
#define UPT_R4 12
#define UPT_R5 8
__global__ void __launch_bounds__(256) k_bloom_up_tail(BloomLevels L) {
    __shared__ uint32_t l3[UPT_R3 * UPT_R3], l4[UPT_R4 * UPT_R4], l5[UPT_R5 * UPT_R5];
    const int bx = blockIdx.x, by = blockIdx.y;
    const int o2x = UPT_T2 * bx, o2y = UPT_T2 * by, o3x = 16 * bx - 1, o3y = 16 * by - 1, o4x = 8 * bx - 2, o4y = 8 * by - 2, o5x = 4 * bx - 2, o5y = 4 * by - 2;
    if (threadIdx.x < UPT_R5 * UPT_R5) {
        const int ly = threadIdx.x / UPT_R5, lx = threadIdx.x % UPT_R5;
        l5[threadIdx.x] = L.mip[5][(size_t)clampi(o5y + ly, 0, L.h[5] - 1) * L.w[5] + clampi(o5x + lx, 0, L.w[5] - 1)];
    }
    if (threadIdx.x < UPT_R4 * UPT_R4) {
        const int ly = threadIdx.x / UPT_R4, lx = threadIdx.x % UPT_R4;
        l4[threadIdx.x] = L.mip[4][(size_t)clampi(o4y + ly, 0, L.h[4] - 1) * L.w[4] + clampi(o4x + lx, 0, L.w[4] - 1)];
    }
    for (int i = threadIdx.x; i < UPT_R3 * UPT_R3; i += 256) {
        const int ly = i / UPT_R3, lx = i - ly * UPT_R3;
        l3[i] = L.mip[3][(size_t)clampi(o3y + ly, 0, L.h[3] - 1) * L.w[3] + clampi(o3x + lx, 0, L.w[3] - 1)];
    }
    __syncthreads();
    // mip 4 += tent(mip 5)
    uint32_t v4 = 0u;
    int c4x = 0, c4y = 0;
    bool in4 = false;
    if (threadIdx.x < UPT_R4 * UPT_R4) {
        c4x = o4x + (int)(threadIdx.x % UPT_R4); c4y = o4y + (int)(threadIdx.x / UPT_R4);
        in4 = c4x >= 0 && c4y >= 0 && c4x < L.w[4] && c4y < L.h[4];
        if (in4) v4 = tent_lds(l5, UPT_R5, o5x, o5y, L.w[5], L.h[5], c4x >> 1, c4y >> 1, l4[threadIdx.x]);
    }
    __syncthreads();
    if (threadIdx.x < UPT_R4 * UPT_R4 && in4) {
        l4[threadIdx.x] = v4;
        if (c4x >= 8 * bx && c4x < 8 * bx + 8 && c4y >= 8 * by && c4y < 8 * by + 8) L.fin4[(size_t)c4y * L.w[4] + c4x] = v4;
    }
    __syncthreads();
    // mip 3 += tent(mip 4)
    uint32_t v3[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = (int)threadIdx.x + 256 * k;
        if (i < UPT_R3 * UPT_R3) {
            const int cx = o3x + i % UPT_R3, cy = o3y + i / UPT_R3;
            if (cx >= 0 && cy >= 0 && cx < L.w[3] && cy < L.h[3]) v3[k] = tent_lds(l4, UPT_R4, o4x, o4y, L.w[4], L.h[4], cx >> 1, cy >> 1, l3[i]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = (int)threadIdx.x + 256 * k;
        if (i < UPT_R3 * UPT_R3) {
            const int cx = o3x + i % UPT_R3, cy = o3y + i / UPT_R3;
            if (cx >= 0 && cy >= 0 && cx < L.w[3] && cy < L.h[3]) {
                l3[i] = v3[k];
                if (cx >= 16 * bx && cx < 16 * bx + 16 && cy >= 16 * by && cy < 16 * by + 16) L.fin3[(size_t)cy * L.w[3] + cx] = v3[k];
            }
        }
    }
    __syncthreads();
    // mip 2 += tent(mip 3): four texels per thread
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = (int)threadIdx.x + 256 * k, x = o2x + (i & (UPT_T2 - 1)), y = o2y + (i / UPT_T2);
        if (x < L.w[2] && y < L.h[2]) {
            uint32_t* p = L.mip[2] + (size_t)y * L.w[2] + x;
            *p = tent_lds(l3, UPT_R3, o3x, o3y, L.w[3], L.h[3], x >> 1, y >> 1, *p);
        }
    }
}

// tonemap.comp:26-28; the two divisions feed a continuous curve, the hardware reciprocal is enough
RDR_DEV float tone(float x) {
#pragma clang fp contract(fast)
    const float shoulderStr = 0.22f, linStr = 0.30f, linAngle = 0.10f, toeStr = 0.20f, toeNum = 0.01f, toeDenom = 0.30f;
    return ((x * (shoulderStr * x + linAngle * linStr) + toeStr * toeNum) * __builtin_amdgcn_rcpf(x * (shoulderStr * x + linStr) + toeStr * toeDenom)) - toeNum / toeDenom;
}

// tonemap.comp:45-50 for one channel. exp and pow go straight to the hardware exp2 / log2: the argument of the
// first is <= 0 and the second sees [0, 1), so the library's range fix-ups buy nothing at 8-bit output precision.
// `scale2` = 2^exposure * log2(e).
RDR_DEV float tonemap_value(float c, float bloom, float scale2, float inv_gamma, float inv_white) {
#pragma clang fp contract(fast)
    float v = c * (1.0f - 0.04f) + bloom * 0.04f;
    v = tone(v) * inv_white;
    v = 1.0f - __builtin_amdgcn_exp2f(-v * scale2);
    return __builtin_amdgcn_exp2f(inv_gamma * __builtin_amdgcn_logf(v));
}
RDR_DEV uint32_t tonemap_channel(float c, float bloom, float scale2, float inv_gamma, float inv_white) {
#pragma clang fp contract(fast)
    const float v = tonemap_value(c, bloom, scale2, inv_gamma, inv_white);
    return (uint32_t)(int)(fminf(fmaxf(v, 0.0f), 1.0f) * 255.0f + 0.5f);  // UNORM8, NaN -> 0
}

// the nine tent taps around (ux, uy) as staged texels (binary16 values, see stage_from_mip): the fma chain converts them
// on the fly
struct TentH { uint32_t rg[9], b[9]; };
RDR_DEV void load_tent_h(const uint32_t* __restrict__ src, int sw, int sh, int ux, int uy, TentH& t) {
    const int dxs[9] = {-1, 0, +1, -1, 0, +1, -1, 0, +1};
    const int dys[9] = {-1, -1, -1, 0, 0, 0, +1, +1, +1};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int sx = clampi(ux + dxs[k], 0, sw - 1), sy = clampi(uy + dys[k], 0, sh - 1);
        const uint2 v = stage_from_mip(src[(size_t)sy * sw + sx]);
        t.rg[k] = v.x; t.b[k] = v.y;
    }
}
RDR_DEV void tent_accumulate_h(const TentH& t, float& acc0, float& acc1, float& acc2) {
    const float ws[9] = {0.0625f, 0.125f, 0.0625f, 0.125f, 0.25f, 0.125f, 0.0625f, 0.125f, 0.0625f};
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const rdr_f16x2 rg = as_f16x2(t.rg[k]), bb = as_f16x2(t.b[k]);
        acc0 = fmaf(ws[k], (float)rg.x, acc0); acc1 = fmaf(ws[k], (float)rg.y, acc1); acc2 = fmaf(ws[k], (float)bb.x, acc2);
    }
}

// Last up-sample (mip 1 -> mip 0, destination read from the colour image) + tonemap.comp:34-53 for a 2x2 quad.
// `inv_white` = 1 / tone(11.2) from the host. VEC: the image width is even, so the two pixels of a quad row are
// one 16-byte load and 8-byte stores. KEEP: also store the accumulated mip 0 and the whole log-luminance image (the
// shader writes both; nothing later in the frame reads mip 0, and the average-luminance pass reads only the centre
// crop [cx0, cx1] x [cy0, cy1] of the log-luminance, which is always stored). HDR16: the output image is RGBA16F
// (hdrOutputEnabled, VulkanRenderer.cpp:668-673): `out` then holds 8 bytes per pixel, the same values unquantised.
template <bool VEC, bool KEEP, bool HDR16>
__global__ void __launch_bounds__(256) k_tonemap(const uint2* __restrict__ hdr, const uint32_t* __restrict__ mip1, int sw, int sh,
                                                 uint32_t* __restrict__ bloom0, uint32_t* __restrict__ out, float* __restrict__ loglum,
                                                 int W, int H, float scale2, float inv_gamma, float inv_white, float log2_6, int cx0, int cx1,
                                                 int cy0, int cy1) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63), qy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (2 * qx >= W || 2 * qy >= H) return;
    TentH t;
    const bool have_mip1 = sw > 0 && sh > 0;
    if (have_mip1) load_tent_h(mip1, sw, sh, qx, qy, t);
#pragma unroll
    for (int oy = 0; oy < 2; oy++) {
        const int y = 2 * qy + oy;
        if (y >= H) continue;
        const size_t i0 = (size_t)y * W + 2 * qx;
        uint2 h[2];
        if (VEC) { const uint4 v = *(const uint4*)(hdr + i0); h[0] = make_uint2(v.x, v.y); h[1] = make_uint2(v.z, v.w); }
        else { h[0] = hdr[i0]; h[1] = (2 * qx + 1 < W) ? hdr[i0 + 1] : make_uint2(0u, 0u); }
        uint32_t b0[2] = {0u, 0u}, px[2];
        uint2 hp[2];
        float ll[2];
#pragma unroll
        for (int ox = 0; ox < 2; ox++) {
            float c[3], a, bl[3];
            unpack_half4(h[ox], c[0], c[1], c[2], a);
            // mip 0 = the colour truncated to the packed format (a mask on its half bits), plus the tent of mip 1
            const uint2 m0 = stage_from_hdr(h[ox]);
            uint32_t trg = m0.x, tb = m0.y;  // mip 0 as half bits: red | green << 16, blue
            if (have_mip1) {
                const rdr_f16x2 rg = as_f16x2(m0.x), bb = as_f16x2(m0.y);
                float a0 = (float)rg.x, a1 = (float)rg.y, a2 = (float)bb.x;
                tent_accumulate_h(t, a0, a1, a2);
                // the accumulated value as the packed format stores it. The sums are never negative, so the truncation is
                // a round-toward-zero conversion to half and a mask; NaNs stay NaNs (quiet bit above the mask)
                if (KEEP) b0[ox] = pack_r11g11b10(a0, a1, a2);
                trg = half2_bits_rtz(a0, a1) & 0xFFF0FFF0u; tb = half2_bits_rtz(a2, 0.0f) & 0xFFE0u;
            } else if (KEEP) {
                b0[ox] = (m0.x >> 4 & 0x7FFu) | (m0.x >> 9 & 0x3FF800u) | (m0.y >> 5 << 22);
            }
            {
                const rdr_f16x2 rg = as_f16x2(trg), bb = as_f16x2(tb);
                bl[0] = (float)rg.x; bl[1] = (float)rg.y; bl[2] = (float)bb.x;
            }
            ll[ox] = __builtin_amdgcn_logf((2.0f * c[0] + 3.0f * c[1]) + c[2]) - log2_6;
            if (HDR16)
                hp[ox] = pack_half4(tonemap_value(c[0], bl[0], scale2, inv_gamma, inv_white), tonemap_value(c[1], bl[1], scale2, inv_gamma, inv_white),
                                    tonemap_value(c[2], bl[2], scale2, inv_gamma, inv_white), 1.0f);
            else
                px[ox] = 0xFF000000u | tonemap_channel(c[0], bl[0], scale2, inv_gamma, inv_white) |
                         (tonemap_channel(c[1], bl[1], scale2, inv_gamma, inv_white) << 8) |
                         (tonemap_channel(c[2], bl[2], scale2, inv_gamma, inv_white) << 16);
        }
        const bool lum = KEEP || (y >= cy0 && y <= cy1 && 2 * qx + 1 >= cx0 && 2 * qx <= cx1);
        if (VEC) {
            if (KEEP) *(uint2*)(bloom0 + i0) = make_uint2(b0[0], b0[1]);
            if (HDR16) *(uint4*)((uint2*)out + i0) = make_uint4(hp[0].x, hp[0].y, hp[1].x, hp[1].y);
            else *(uint2*)(out + i0) = make_uint2(px[0], px[1]);
            if (lum) *(float2*)(loglum + i0) = make_float2(ll[0], ll[1]);
        } else {
            if (HDR16) ((uint2*)out)[i0] = hp[0];
            else out[i0] = px[0];
            if (KEEP) bloom0[i0] = b0[0];
            if (lum) loglum[i0] = ll[0];
            if (2 * qx + 1 < W) {
                if (HDR16) ((uint2*)out)[i0 + 1] = hp[1];
                else out[i0 + 1] = px[1];
                if (KEEP) bloom0[i0 + 1] = b0[1];
                if (lum) loglum[i0 + 1] = ll[1];
            }
        }
    }
}

// averageLuminance.comp on a 2N x 2N input -> N/8 x N/8 groups, one block per
// group, 64 threads, each fetching a 2x2 quad; the tree adds in the shader's
// order. `crop` != 0: the input is the W x H log-luminance image and texel
// (x,y) is read at the centre-crop offset with clamp-to-edge
// (VulkanRenderer.cpp:1165).
__global__ void __launch_bounds__(64) k_lum_reduce(const float* __restrict__ in, int in_w, int in_h, int ox, int oy, int dim,
                                                   float* __restrict__ out, int groups) {
    __shared__ float a[8][8];  // localAverages[x][y]
    const int lx = threadIdx.x & 7, ly = threadIdx.x >> 3;
    const int gx = blockIdx.x % groups, gy = blockIdx.x / groups;
    const int sx = (gx * 8 + lx) * 2, sy = (gy * 8 + ly) * 2;
    const float f = 1.0f / (8 * 8 * 4);
    float l[4];
    const int offx[4] = {0, 0, 1, 1}, offy[4] = {0, 1, 0, 1};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int x = sx + offx[k], y = sy + offy[k];
        if (dim == 0) { x = clampi(ox + x, 0, in_w - 1); y = clampi(oy + y, 0, in_h - 1); }
        l[k] = in[(size_t)y * in_w + x];
    }
    a[lx][ly] = ((l[0] * f + l[1] * f) + l[2] * f) + l[3] * f;
    __syncthreads();
    for (int i = 4; i > 0; i >>= 1) {
        float v = 0.0f;
        const bool act = lx < i && ly < i;
        if (act) v = a[lx][ly] + ((a[lx + i][ly + i] + a[lx][ly + i]) + a[lx + i][ly]);
        __syncthreads();
        if (act) a[lx][ly] = v;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = a[0][0];
}

}  // namespace

void launch_bloom_down(hipStream_t s, const void* src, bool src_is_hdr, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    if (src_is_hdr) hipLaunchKernelGGL(k_bloom_down<true>, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
    else hipLaunchKernelGGL(k_bloom_down<false>, grid, dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
void launch_bloom_up(hipStream_t s, const uint32_t* src, int sw, int sh, uint32_t* dst, int dw, int dh, int dst_pitch) {
    if (dw <= 0 || dh <= 0) return;
    const int qw = (dw + 1) / 2, qh = (dh + 1) / 2;
    hipLaunchKernelGGL(k_bloom_up, dim3((qw + 63) / 64, (qh + 3) / 4), dim3(256), 0, s, src, sw, sh, dst, dw, dh, dst_pitch);
}
// Everything of the bloom chain that precedes the tonemap kernel: five down-samples, then the up-samples down to
// mip 1 (the last one, into mip 0, is part of the tonemap kernel). mip[k] / w[k] / h[k]: the six levels, level k of
// size max(1, W >> k) x max(1, H >> k), rows packed. Images with at least 128 pixels a side take the staged kernels
// (LDS tiles, the small levels fused); smaller ones, where a level can vanish, the plain per-level passes. Returns true
// when the staged kernels ran: the accumulated mips 3 and 4 are then in fin3 / fin4 (w[3] x h[3] and w[4] x h[4]
// texels) and mip[3] / mip[4] keep the down-sampled values; otherwise they were accumulated in place.
bool launch_bloom_chain(hipStream_t s, const void* hdr, uint32_t* const mip[6], const int w[6], const int h[6], int W, int H, uint32_t* fin3,
                        uint32_t* fin4) {
    if (W >= 128 && H >= 128) {
        BloomLevels L;
        for (int k = 0; k < 6; k++) { L.mip[k] = mip[k]; L.w[k] = w[k]; L.h[k] = h[k]; }
        L.fin3 = fin3; L.fin4 = fin4;
        hipLaunchKernelGGL(k_bloom_down_lds<true>, dim3((w[1] + DOWN_TX - 1) / DOWN_TX, (h[1] + DOWN_TY - 1) / DOWN_TY), dim3(256), 0, s, hdr, w[0], h[0],
                           mip[1], w[1], h[1], w[1]);
        hipLaunchKernelGGL(k_bloom_down_lds<false>, dim3((w[2] + DOWN_TX - 1) / DOWN_TX, (h[2] + DOWN_TY - 1) / DOWN_TY), dim3(256), 0, s, (const void*)mip[1],
                           w[1], h[1], mip[2], w[2], h[2], w[2]);
        // sizes round down from level to level, so mip 3 can reach past 4x the extent of mip 5: the grid covers all three
        const int gx = max(max((w[5] + TAIL_T5 - 1) / TAIL_T5, (w[4] + 2 * TAIL_T5 - 1) / (2 * TAIL_T5)), (w[3] + 4 * TAIL_T5 - 1) / (4 * TAIL_T5));
        const int gy = max(max((h[5] + TAIL_T5 - 1) / TAIL_T5, (h[4] + 2 * TAIL_T5 - 1) / (2 * TAIL_T5)), (h[3] + 4 * TAIL_T5 - 1) / (4 * TAIL_T5));
        hipLaunchKernelGGL(k_bloom_down_tail, dim3(gx, gy), dim3(256), 0, s, L);
        hipLaunchKernelGGL(k_bloom_up_tail, dim3((w[2] + UPT_T2 - 1) / UPT_T2, (h[2] + UPT_T2 - 1) / UPT_T2), dim3(256), 0, s, L);
        launch_bloom_up(s, mip[2], w[2], h[2], mip[1], w[1], h[1], w[1]);
        return true;
    }
    for (int i = 0; i < 5; i++)
        launch_bloom_down(s, i == 0 ? hdr : (const void*)mip[i], i == 0, w[i], h[i], mip[i + 1], W >> (i + 1), H >> (i + 1), w[i + 1]);
    for (int i = 0; i < 4; i++) {
        const int dl = 4 - i, sl = 5 - i;
        launch_bloom_up(s, mip[sl], w[sl], h[sl], mip[dl], W >> dl, H >> dl, w[dl]);
    }
    return false;
}
void launch_tonemap(hipStream_t s, const void* hdr, const uint32_t* mip1, int sw, int sh, uint32_t* bloom0, uint32_t* out, float* loglum,
                    int W, int H, float scale, float inv_gamma, float white, float log2_6, bool keep, bool hdr16) {
    const int qw = (W + 1) / 2, qh = (H + 1) / 2;
    const float scale2 = scale * 1.44269504088896340736f;  // exp(-x s) = exp2(-x s log2 e)
    const dim3 grid((qw + 63) / 64, (qh + 3) / 4);
    // what launch_average_luminance reads: 256 x 256 texels from (W/2 - 128, H/2 - 128), coordinates clamped to the image
    const int cx0 = max(W / 2 - 128, 0), cx1 = min(W / 2 + 127, W - 1), cy0 = max(H / 2 - 128, 0), cy1 = min(H / 2 + 127, H - 1);
#define RDR_TONEMAP(VEC, KEEP, HDR16)                                                                                                          \
    hipLaunchKernelGGL((k_tonemap<VEC, KEEP, HDR16>), grid, dim3(256), 0, s, (const uint2*)hdr, mip1, sw, sh, bloom0, out, loglum, W, H, scale2, \
                       inv_gamma, 1.0f / white, log2_6, cx0, cx1, cy0, cy1)
#define RDR_TONEMAP_OUT(VEC, KEEP) do { if (hdr16) RDR_TONEMAP(VEC, KEEP, true); else RDR_TONEMAP(VEC, KEEP, false); } while (0)
    if ((W & 1) == 0) { if (keep) RDR_TONEMAP_OUT(true, true); else RDR_TONEMAP_OUT(true, false); }
    else { if (keep) RDR_TONEMAP_OUT(false, true); else RDR_TONEMAP_OUT(false, false); }
#undef RDR_TONEMAP_OUT
#undef RDR_TONEMAP
}
void launch_average_luminance(hipStream_t s, const float* loglum, int W, int H, float* l16, float* l1) {
    hipLaunchKernelGGL(k_lum_reduce, dim3(256), dim3(64), 0, s, loglum, W, H, W / 2 - 128, H / 2 - 128, 0, l16, 16);
    hipLaunchKernelGGL(k_lum_reduce, dim3(1), dim3(64), 0, s, (const float*)l16, 16, 16, 0, 0, 16, l1, 1);
}
