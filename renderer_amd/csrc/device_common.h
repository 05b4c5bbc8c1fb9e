// Device-side helpers shared by the kernels: storage-format conversions,
// exact transforms, texture/cube/shadow sampling.
//
// Files that include this header are compiled with -ffp-contract=off: every
// fused multiply-add is written out as fmaf()/fma(), and float '/' and sqrtf
// are IEEE correctly rounded (hipcc default). That makes the values that feed
// discontinuous decisions (coverage, depth, shadow compares) independent of
// the compiler's instruction selection.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rdr_device.h"

#define RDR_DEV __device__ __forceinline__

// Pointers that kernels find inside structures in memory are generic to the compiler, and loads through them
// become flat_load (which also ties up the LDS counter). Everything such a pointer can point at here is HBM.
#define RDR_GLOBAL __attribute__((address_space(1)))
template <class T> RDR_DEV const RDR_GLOBAL T* as_global(const T* p) { return (const RDR_GLOBAL T*)p; }
// Two consecutive dwords at a 4-byte-aligned address in one load (the hardware takes multi-dword global loads at
// dword alignment): halves the number of load instructions of a 2x2 filter footprint.
typedef uint32_t rdr_uint2_a4 __attribute__((ext_vector_type(2), aligned(4)));
RDR_DEV void load_pair(const RDR_GLOBAL uint32_t* p, uint32_t& a, uint32_t& b) {
    const rdr_uint2_a4 v = *(const RDR_GLOBAL rdr_uint2_a4*)p;
    a = v.x; b = v.y;
}

struct F3 { float x, y, z; };
struct F4 { float x, y, z, w; };

RDR_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
RDR_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
RDR_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
RDR_DEV F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
RDR_DEV float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RDR_DEV F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
// v * (1 / sqrt(dot(v,v))) with correctly rounded sqrt and divide
RDR_DEV F3 normalize3(F3 a) { const float s = 1.0f / sqrtf(dot3(a, a)); return a * s; }
RDR_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// m is column-major: m[col*4+row]
RDR_DEV F4 xform_point(const float* m, float x, float y, float z) {
    F4 r;
    r.x = fmaf(m[0], x, fmaf(m[4], y, fmaf(m[8], z, m[12])));
    r.y = fmaf(m[1], x, fmaf(m[5], y, fmaf(m[9], z, m[13])));
    r.z = fmaf(m[2], x, fmaf(m[6], y, fmaf(m[10], z, m[14])));
    r.w = fmaf(m[3], x, fmaf(m[7], y, fmaf(m[11], z, m[15])));
    return r;
}
RDR_DEV F3 xform_dir(const float* m, float x, float y, float z) {
    return f3(fmaf(m[0], x, fmaf(m[4], y, m[8] * z)), fmaf(m[1], x, fmaf(m[5], y, m[9] * z)), fmaf(m[2], x, fmaf(m[6], y, m[10] * z)));
}

// ---- float16 (colour attachment R16G16B16A16_SFLOAT), round to nearest even
RDR_DEV uint32_t f32_to_f16_bits(float f) { const _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return (uint32_t)u; }
RDR_DEV float f16_bits_to_f32(uint32_t b) { const unsigned short u = (unsigned short)b; _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }
RDR_DEV float quantize_half(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }
RDR_DEV uint2 pack_half4(float r, float g, float b, float a) {
    return make_uint2(f32_to_f16_bits(r) | (f32_to_f16_bits(g) << 16), f32_to_f16_bits(b) | (f32_to_f16_bits(a) << 16));
}
RDR_DEV void unpack_half4(uint2 p, float& r, float& g, float& b, float& a) {
    r = f16_bits_to_f32(p.x & 0xFFFFu); g = f16_bits_to_f32(p.x >> 16);
    b = f16_bits_to_f32(p.y & 0xFFFFu); a = f16_bits_to_f32(p.y >> 16);
}

// ---- B10G11R11_UFLOAT: 5 exponent bits, MB mantissa bits, no sign. Negative
// -> 0, NaN -> NaN, +inf -> inf, above the largest finite -> largest finite,
// mantissa truncated, denormals kept.
// The format has binary16's exponent field, so it is binary16 with the sign
// and the low 10 - MB mantissa bits dropped: encoding is one
// round-toward-zero conversion to half (which also saturates finite overflow
// to 65504, whose top bits are the largest finite packed value) and a shift;
// decoding is a shift and a half -> float conversion.
typedef __fp16 rdr_half2_t __attribute__((ext_vector_type(2)));
RDR_DEV uint32_t half2_bits_rtz(float a, float b) {
    const rdr_half2_t h = __builtin_amdgcn_cvt_pkrtz(a, b);
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}
template <int MB> RDR_DEV uint32_t ufloat_from_half_bits(uint32_t h) {  // h: one binary16 in the low 16 bits
    const uint32_t mag = h & 0x7FFFu;
    if (mag > 0x7C00u) return (31u << MB) | 1u;  // NaN of either sign
    return (h & 0x8000u) ? 0u : mag >> (10 - MB);
}
template <int MB> RDR_DEV float ufloat_to_float(uint32_t v) { return f16_bits_to_f32(v << (10 - MB)); }
RDR_DEV uint32_t pack_r11g11b10(float r, float g, float b) {
    const uint32_t rg = half2_bits_rtz(r, g), b0 = half2_bits_rtz(b, 0.0f);
    return ufloat_from_half_bits<6>(rg & 0xFFFFu) | (ufloat_from_half_bits<6>(rg >> 16) << 11) | (ufloat_from_half_bits<5>(b0 & 0xFFFFu) << 22);
}
RDR_DEV void unpack_r11g11b10(uint32_t p, float& r, float& g, float& b) {
    r = ufloat_to_float<6>(p & 0x7FFu);
    g = ufloat_to_float<6>((p >> 11) & 0x7FFu);
    b = ufloat_to_float<5>(p >> 22);
}

// UNORM8: clamp, scale, round half up; NaN -> 0
RDR_DEV uint32_t float_to_unorm8(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 1.0f) return 255u;
    return (uint32_t)(int)(f * 255.0f + 0.5f);
}

// ---- cube maps: Vulkan face selection (z wins ties over y over x), bilinear
// inside the face with clamp-to-edge
RDR_DEV int cube_face(float rx, float ry, float rz, float& s, float& t) {
    const float ax = fabsf(rx), ay = fabsf(ry), az = fabsf(rz);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { face = rz < 0.0f ? 5 : 4; sc = rz < 0.0f ? -rx : rx; tc = -ry; ma = az; }
    else if (ay >= ax) { face = ry < 0.0f ? 3 : 2; sc = rx; tc = ry < 0.0f ? -rz : rz; ma = ay; }
    else { face = rx < 0.0f ? 1 : 0; sc = rx < 0.0f ? rz : -rz; tc = -ry; ma = ax; }
    s = 0.5f * (sc / ma) + 0.5f;
    t = 0.5f * (tc / ma) + 0.5f;
    return face;
}
// Texel (i, j) of `face` at resolution r, where i or j may be one step outside [0, r): the texel across the cube
// edge. Vulkan filters linearly across cube faces (seamless cube maps are the default), which makes a lookup
// continuous in its direction. Texel centres are odd integers on a grid of half-texel steps, U = 2i + 1 - r in
// (-r, r) on a face at distance r; a step outside is folded over the edge (distance r - 1 on the old major axis,
// r on the crossed one), so the along-edge index carries over exactly. At a corner (both outside) the t index is
// clamped first. Integer arithmetic only.
RDR_DEV void cube_fold(int face, int i, int j, int r, int& face_out, int& i_out, int& j_out) {
    if (i >= 0 && i < r && j >= 0 && j < r) { face_out = face; i_out = i; j_out = j; return; }
    if ((i < 0 || i >= r) && (j < 0 || j >= r)) j = j < 0 ? 0 : r - 1;
    int U = 2 * i + 1 - r, V = 2 * j + 1 - r, M = r;
    if (U > r) { M = 2 * r - U; U = r; } else if (U < -r) { M = 2 * r + U; U = -r; }
    else if (V > r) { M = 2 * r - V; V = r; } else if (V < -r) { M = 2 * r + V; V = -r; }
    int X, Y, Z;  // point on the cube, components in half-texel units
    switch (face) {
        case 0: X = M; Y = -V; Z = -U; break;
        case 1: X = -M; Y = -V; Z = U; break;
        case 2: X = U; Y = M; Z = V; break;
        case 3: X = U; Y = -M; Z = -V; break;
        case 4: X = U; Y = -V; Z = M; break;
        default: X = -U; Y = -V; Z = -M; break;
    }
    const int ax = X < 0 ? -X : X, ay = Y < 0 ? -Y : Y, az = Z < 0 ? -Z : Z;
    int sc, tc;
    if (az >= ax && az >= ay) { face_out = Z < 0 ? 5 : 4; sc = Z < 0 ? -X : X; tc = -Y; }
    else if (ay >= ax) { face_out = Y < 0 ? 3 : 2; sc = X; tc = Y < 0 ? -Z : Z; }
    else { face_out = X < 0 ? 1 : 0; sc = X < 0 ? Z : -Z; tc = -Y; }
    i_out = (sc + r - 1) / 2;
    j_out = (tc + r - 1) / 2;
    i_out = i_out < 0 ? 0 : (i_out > r - 1 ? r - 1 : i_out);
    j_out = j_out < 0 ? 0 : (j_out > r - 1 ? r - 1 : j_out);
}
RDR_DEV uint32_t cube_texel_offset(int face, int i, int j, int r) {
    int f2, i2, j2;
    cube_fold(face, i, j, r, f2, i2, j2);
    return (uint32_t)((f2 * r + j2) * r + i2);
}
// Offsets of the 2x2 footprint at (i0, j0), i0/j0 in [-1, r-1]. The common case (all four inside the face) costs
// four adds; only footprints that touch the border of a face go through the fold.
RDR_DEV void cube_footprint(int face, int i0, int j0, int r, uint32_t o[4]) {
    if (i0 >= 0 && j0 >= 0 && i0 + 1 < r && j0 + 1 < r) {
        const uint32_t base = (uint32_t)((face * r + j0) * r + i0);
        o[0] = base; o[1] = base + 1u; o[2] = base + (uint32_t)r; o[3] = base + (uint32_t)r + 1u;
    } else {
        o[0] = cube_texel_offset(face, i0, j0, r); o[1] = cube_texel_offset(face, i0 + 1, j0, r);
        o[2] = cube_texel_offset(face, i0, j0 + 1, r); o[3] = cube_texel_offset(face, i0 + 1, j0 + 1, r);
    }
}

// Bit-exact variant (IEEE divides, weights as (1-a)*x + a*y): used by the bakes, whose outputs are compared
// bit for bit with the CPU checker.
RDR_DEV void cube_sample(const DCube& cube, uint32_t level, float dx, float dy, float dz, float out[3]) {
    float s, t;
    const int face = cube_face(dx, dy, dz, s, t);
    const uint32_t rr = cube.res >> level;
    const int r = (int)(rr ? rr : 1u);
    const RDR_GLOBAL uint32_t* px = as_global(cube.texels) + cube.level_offset[level];
    const float tu = s * (float)r - 0.5f, tv = t * (float)r - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, -1), r - 1), j0 = min(max((int)fv, -1), r - 1);
    uint32_t o[4];
    cube_footprint(face, i0, j0, r, o);
    float c00[3], c10[3], c01[3], c11[3];
    unpack_r11g11b10(px[o[0]], c00[0], c00[1], c00[2]);
    unpack_r11g11b10(px[o[1]], c10[0], c10[1], c10[2]);
    unpack_r11g11b10(px[o[2]], c01[0], c01[1], c01[2]);
    unpack_r11g11b10(px[o[3]], c11[0], c11[1], c11[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float top = (1.0f - a) * c00[k] + a * c10[k], bot = (1.0f - a) * c01[k] + a * c11[k];
        out[k] = (1.0f - b) * top + b * bot;
    }
}

// cube_sample on a bordered copy of the cube (every face (r + 2)^2, the border holding the texels across the cube's
// edges, k_cube_border): the footprint is four neighbouring words wherever it lies, the arithmetic is cube_sample's.
RDR_DEV void cube_sample_bordered(const DCube& cube, uint32_t level, float dx, float dy, float dz, float out[3]) {
    float s, t;
    const int face = cube_face(dx, dy, dz, s, t);
    const uint32_t rr = cube.res >> level;
    const int r = (int)(rr ? rr : 1u), pitch = r + 2;
    const RDR_GLOBAL uint32_t* px = as_global(cube.texels) + cube.level_offset[level];
    const float tu = s * (float)r - 0.5f, tv = t * (float)r - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, -1), r - 1), j0 = min(max((int)fv, -1), r - 1);
    const RDR_GLOBAL uint32_t* row = px + ((face * pitch + j0 + 1) * pitch + i0 + 1);
    float c00[3], c10[3], c01[3], c11[3];
    unpack_r11g11b10(row[0], c00[0], c00[1], c00[2]);
    unpack_r11g11b10(row[1], c10[0], c10[1], c10[2]);
    unpack_r11g11b10(row[pitch], c01[0], c01[1], c01[2]);
    unpack_r11g11b10(row[pitch + 1], c11[0], c11[1], c11[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float top = (1.0f - a) * c00[k] + a * c10[k], bot = (1.0f - a) * c01[k] + a * c11[k];
        out[k] = (1.0f - b) * top + b * bot;
    }
}

// Products of two values below 2^23 (image sizes, rows, faces): the 24-bit multiplier issues at full rate, the 32-bit
// one at a quarter of it.
RDR_DEV int mul24(int a, int b) { return __mul24(a, b); }
// Variants for shading, where the result is continuous in the direction (filtering is seamless across cube edges, so
// which of two tied faces is chosen does not show): the hardware's cube-map instructions select the face by the same
// rule as cube_face above (z wins ties over y over x) and return sc, tc and twice the major axis; the division is
// the hardware reciprocal.
RDR_DEV int cube_face_fast(float rx, float ry, float rz, float& s, float& t) {
#pragma clang fp contract(fast)
    const float sc = __builtin_amdgcn_cubesc(rx, ry, rz), tc = __builtin_amdgcn_cubetc(rx, ry, rz);
    const float hinv = __builtin_amdgcn_rcpf(fabsf(__builtin_amdgcn_cubema(rx, ry, rz)));  // 1 / (2 |major axis|)
    s = fminf(fmaxf(fmaf(sc, hinv, 0.5f), 0.0f), 1.0f);
    t = fminf(fmaxf(fmaf(tc, hinv, 0.5f), 0.0f), 1.0f);
    return (int)__builtin_amdgcn_cubeid(rx, ry, rz);
}
// `cube` is a bordered copy (k_cube_border): the footprint at (i0, j0), i0 and j0 in [-1, r - 1], is two pairs of
// neighbouring words, no edge handling.
RDR_DEV void cube_sample_fast(const DCube& cube, uint32_t level, float dx, float dy, float dz, float out[3]) {
#pragma clang fp contract(fast)
    float s, t;
    const int face = cube_face_fast(dx, dy, dz, s, t);
    const uint32_t rr = cube.res >> level;
    const int r = (int)(rr ? rr : 1u), pitch = r + 2;
    const RDR_GLOBAL uint32_t* px = as_global(cube.texels) + cube.level_offset[level];
    const float tu = s * (float)r - 0.5f, tv = t * (float)r - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, -1), r - 1), j0 = min(max((int)fv, -1), r - 1);
    const RDR_GLOBAL uint32_t* row = px + (mul24(mul24(face, pitch) + j0 + 1, pitch) + i0 + 1);
    uint32_t q00, q10, q01, q11;
    load_pair(row, q00, q10);
    load_pair(row + pitch, q01, q11);
    float c00[3], c10[3], c01[3], c11[3];
    unpack_r11g11b10(q00, c00[0], c00[1], c00[2]);
    unpack_r11g11b10(q10, c10[0], c10[1], c10[2]);
    unpack_r11g11b10(q01, c01[0], c01[1], c01[2]);
    unpack_r11g11b10(q11, c11[0], c11[1], c11[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float top = fmaf(a, c10[k] - c00[k], c00[k]), bot = fmaf(a, c11[k] - c01[k], c01[k]);
        out[k] = fmaf(b, bot - top, top);
    }
}

// `cube` is a footprint-major copy (k_cube_quads): the whole 2x2 footprint at (i0, j0) is one 16-byte entry.
typedef uint32_t rdr_uint4_t __attribute__((ext_vector_type(4)));
RDR_DEV void cube_sample_quad(const DCube& cube, uint32_t level, float dx, float dy, float dz, float out[3]) {
#pragma clang fp contract(fast)
    float s, t;
    const int face = cube_face_fast(dx, dy, dz, s, t);
    const uint32_t rr = cube.res >> level;
    const int r = (int)(rr ? rr : 1u), pitch = r + 1;
    const RDR_GLOBAL rdr_uint4_t* px = (const RDR_GLOBAL rdr_uint4_t*)as_global(cube.texels) + cube.level_offset[level];
    const float tu = s * (float)r - 0.5f, tv = t * (float)r - 0.5f;
    const float fu = floorf(tu), fv = floorf(tv);
    const float a = tu - fu, b = tv - fv;
    const int i0 = min(max((int)fu, -1), r - 1), j0 = min(max((int)fv, -1), r - 1);
    const rdr_uint4_t q = px[mul24(mul24(face, pitch) + j0 + 1, pitch) + i0 + 1];
    float c00[3], c10[3], c01[3], c11[3];
    unpack_r11g11b10(q.x, c00[0], c00[1], c00[2]);
    unpack_r11g11b10(q.y, c10[0], c10[1], c10[2]);
    unpack_r11g11b10(q.z, c01[0], c01[1], c01[2]);
    unpack_r11g11b10(q.w, c11[0], c11[1], c11[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float top = fmaf(a, c10[k] - c00[k], c00[k]), bot = fmaf(a, c11[k] - c01[k], c01[k]);
        out[k] = fmaf(b, bot - top, top);
    }
}

// ---- 2D material textures: trilinear, isotropic LOD
#define RDR_INV255 (1.0f / 255.0f)

// `srgb` points at the 256-entry sRGB decode table (LDS copy in the hot kernels).
// UNORM channels decode as byte * (1/255). MASK selects the channels the caller
// reads (bit k = channel k); the others are left untouched. `table` is the
// global DTexture table, read only for images whose level offsets are not
// arithmetic (not a power-of-two square).
RDR_DEV uint32_t tex_level_offset(const DTexDesc& t, const DTexture* __restrict__ table, uint32_t level) {
    const uint32_t flags = t.info >> 5;
    if (flags & RDR_TEXD_POW2_SQUARE) {
        // sum_{i<level} 4^(n-i) with w = h = 2^n: 4^(n-level+1) * (4^level - 1) / 3, and (4^level - 1) / 3 is `level`
        // ones in the even bit positions (no division, no multiplication)
        const uint32_t n = 31u - (uint32_t)__clz((int)(t.wh & 0xFFFFu));
        return (0x55555555u & ((1u << (2u * level)) - 1u)) << (2u * (n - level + 1u));
    }
    return as_global(table)[t.info >> 16].level_offset[level];
}
// Texel pair (i0, i0 + 1) and weight along one axis, without an integer modulo. REPEAT works on the fractional part of
// the coordinate. MIRRORED_REPEAT reflects the coordinate into [0, 1] (period 2) and then clamps the pair to the edge:
// the wrapping operation maps texel -1 to 0 and texel n to n - 1, which is what a clamped footprint of the reflected
// coordinate reads, with the two taps and their weights exchanged on the reflected half. The weights differ from the
// unwrapped formula only by rounding, and bilinear filtering is continuous across texel boundaries.
RDR_DEV float tex_reflect(float coord) {
    const float h = coord * 0.5f;
    const float m = (h - floorf(h)) * 2.0f;  // coord modulo 2, in [0, 2]
    return m > 1.0f ? 2.0f - m : m;
}
RDR_DEV void tex_axis(float coord, int n, uint32_t mode, int& i0, int& i1, float& a) {
#pragma clang fp contract(fast)
    if (mode == 0u) coord -= floorf(coord);
    else if (mode == 1u) coord = tex_reflect(coord);
    const float tc = coord * (float)n - 0.5f;
    const float fc = floorf(tc);
    a = tc - fc;
    if (mode == 0u) {
        const int i = (int)fc;  // in [-1, n-1]
        i0 = i < 0 ? n - 1 : i;
        i1 = i + 1 >= n ? 0 : i + 1;
    } else {
        const int i = (int)fmaxf(fminf(fc, (float)n), -1.0f);
        i0 = max(i, 0);
        i1 = min(i + 1, n - 1);
        i0 = min(i0, n - 1);
    }
}
// Where a sample lands in an image of given size, levels and wrap modes: the level pair, and per level the four
// texel offsets (from the start of the image) with the two filter weights. The material's five textures usually
// share all of that, so it is computed once per fragment and reused for every texture whose geometry matches.
struct TexLevelTaps { uint32_t o00, o10, o01, o11; float a, b; };
struct TexTaps {
    TexLevelTaps lv[2];
    float f;        // weight of the second level
    bool two;       // wave-uniform: the second level is fetched
};
// NEAREST within a level (Vulkan 1.3, 16.8 "nearest filtering": i = floor(u) on the unnormalised coordinate, then the
// wrapping operation of the address mode). REPEAT on the fractional part, as in tex_axis.
RDR_DEV int tex_axis_nearest(float coord, int n, uint32_t mode) {
    if (mode == 0u) {
        coord -= floorf(coord);
        const int i = (int)floorf(coord * (float)n);  // in [0, n]: n when the product rounds up
        return i >= n ? 0 : i;
    }
    if (mode == 1u) coord = tex_reflect(coord);
    return (int)fminf(fmaxf(floorf(coord * (float)n), 0.0f), (float)(n - 1));
}
RDR_DEV void tex_level_taps(const DTexDesc& t, const DTexture* __restrict__ table, uint32_t level, float u, float v, bool nearest, TexLevelTaps& o) {
    const uint32_t ww = (t.wh & 0xFFFFu) >> level, hh = (t.wh >> 16) >> level;
    const int w = (int)(ww ? ww : 1u), h = (int)(hh ? hh : 1u);
    const uint32_t base = tex_level_offset(t, table, level);
    const uint32_t flags = t.info >> 5;
    if (nearest) {
        // one texel, written as a footprint of four equal taps with zero weights: the filter returns it exactly
        const int i = tex_axis_nearest(u, w, (flags >> 2) & 3u), j = tex_axis_nearest(v, h, (flags >> 4) & 3u);
        o.o00 = o.o10 = o.o01 = o.o11 = base + (uint32_t)(mul24(j, w) + i);
        o.a = 0.0f; o.b = 0.0f;
        return;
    }
    int i0, i1, j0, j1;
    tex_axis(u, w, (flags >> 2) & 3u, i0, i1, o.a);
    tex_axis(v, h, (flags >> 4) & 3u, j0, j1, o.b);
    const int r0 = mul24(j0, w), r1 = mul24(j1, w);  // sizes are below 2^16
    o.o00 = base + (uint32_t)(r0 + i0); o.o10 = base + (uint32_t)(r0 + i1);
    o.o01 = base + (uint32_t)(r1 + i0); o.o11 = base + (uint32_t)(r1 + i1);
}
// texture(sampler2D, uv) with screen-space uv derivatives; lambda = log2(rho) = 0.5 * log2(rho^2).
// The Sampler's filters (VulkanRenderer.cpp:366-368 hands all three to the driver): magFilter where lambda <= 0,
// minFilter above; mipmap mode NEAREST takes the one level ceil(lambda + 0.5) - 1, LINEAR blends floor(lambda) and the
// next. tex_taps_filtered is the general form; tex_taps branches to it, once per wave, only when a lane's sampler has
// a NEAREST filter, so that the all-LINEAR case pays one test for it.
RDR_DEV void tex_taps_filtered(const DTexDesc& t, const DTexture* __restrict__ table, float u, float v, float dudx, float dvdx, float dudy,
                               float dvdy, uint32_t nst, TexTaps& o) {
#pragma clang fp contract(fast)
    const uint32_t levels = t.info & 31u;
    float lambda = 0.0f;
    bool minified = false;
    {
        const float fw = (float)(t.wh & 0xFFFFu), fh = (float)(t.wh >> 16);
        const float ax = dudx * fw, ay = dvdx * fh, bx = dudy * fw, by = dvdy * fh;
        const float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
        if (rho2 > 1.0f) { minified = true; lambda = fminf(0.5f * __log2f(rho2), (float)(levels - 1u)); }
    }
    if (nst & 4u) lambda = fmaxf(ceilf(lambda + 0.5f) - 1.0f, 0.0f);
    const float fl = floorf(lambda);
    const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, levels - 1u);
    o.f = lambda - fl;
    o.two = __any(o.f > 0.0f) != 0;  // a lane on a whole level blends the next one in with weight 0, which is exact
    const bool nearest = (((minified ? nst >> 1 : nst) & 1u) != 0u);
    tex_level_taps(t, table, l0, u, v, nearest, o.lv[0]);
    if (o.two) tex_level_taps(t, table, l1, u, v, nearest, o.lv[1]);
}
RDR_DEV void tex_taps(const DTexDesc& t, const DTexture* __restrict__ table, float u, float v, float dudx, float dvdx, float dudy, float dvdy,
                      TexTaps& o) {
#pragma clang fp contract(fast)
    const uint32_t nst = (t.info >> (5u + RDR_TEXD_NEAREST_SHIFT)) & 7u;  // bit 0 mag, bit 1 min, bit 2 mipmap mode
    if (__any(nst != 0u)) { tex_taps_filtered(t, table, u, v, dudx, dvdx, dudy, dvdy, nst, o); return; }
    const uint32_t levels = t.info & 31u;
    float lambda = 0.0f;
    if (levels > 1u) {
        const float fw = (float)(t.wh & 0xFFFFu), fh = (float)(t.wh >> 16);
        const float ax = dudx * fw, ay = dvdx * fh, bx = dudy * fw, by = dvdy * fh;
        const float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
        if (rho2 > 1.0f) lambda = fminf(0.5f * __log2f(rho2), (float)(levels - 1u));
    }
    // one branch for the whole wave: minified anywhere -> two levels everywhere (a magnified lane blends level 0
    // with weight 0 for the second level, which is exact)
    o.two = __any(lambda > 0.0f) != 0;
    const float fl = floorf(lambda);
    const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, levels - 1u);
    o.f = lambda - fl;
    tex_level_taps(t, table, l0, u, v, false, o.lv[0]);
    if (o.two) tex_level_taps(t, table, l1, u, v, false, o.lv[1]);
}
// texture geometry = size, level count, level-offset rule, wrap modes, filters and texel type (not the sRGB bit, not the
// table index)
RDR_DEV bool tex_same_geometry(const DTexDesc& x, const DTexDesc& y) {
    return (int)(x.wh == y.wh) & (int)(((x.info ^ y.info) & (31u | (0x3FEu << 5))) == 0u);
}
// Bilinear filter of four raw RGBA8 texels. sRGB channels go through the decode table before filtering; UNORM
// channels are filtered as bytes and scaled by 1/255 once afterwards.
template <int MASK>
RDR_DEV void tex_filter(const float* srgb, bool is_srgb, uint32_t q00, uint32_t q10, uint32_t q01, uint32_t q11, float a, float b, float out[4]) {
    float c00[4], c10[4], c01[4], c11[4];
    if (is_srgb && (MASK & 7)) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!(MASK & (1 << c))) continue;
            c00[c] = srgb[(q00 >> (8 * c)) & 255u]; c10[c] = srgb[(q10 >> (8 * c)) & 255u];
            c01[c] = srgb[(q01 >> (8 * c)) & 255u]; c11[c] = srgb[(q11 >> (8 * c)) & 255u];
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!(MASK & (1 << c))) continue;
            c00[c] = (float)((q00 >> (8 * c)) & 255u); c10[c] = (float)((q10 >> (8 * c)) & 255u);
            c01[c] = (float)((q01 >> (8 * c)) & 255u); c11[c] = (float)((q11 >> (8 * c)) & 255u);
        }
    }
    if (MASK & 8) { c00[3] = (float)(q00 >> 24); c10[3] = (float)(q10 >> 24); c01[3] = (float)(q01 >> 24); c11[3] = (float)(q11 >> 24); }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (!(MASK & (1 << c))) continue;
        const float top = fmaf(a, c10[c] - c00[c], c00[c]), bot = fmaf(a, c11[c] - c01[c], c01[c]);
        const float v = fmaf(b, bot - top, top);
        out[c] = (c == 3 || !is_srgb) ? v * RDR_INV255 : v;
    }
}
// FLOAT_TEXELS: the kernel is the build that also samples SFLOAT images (launched only when a context holds one, so that
// the common build keeps the register count it has without that branch).
template <int MASK, bool FLOAT_TEXELS = false>
RDR_DEV void tex_fetch_level(const float* srgb, const DTexDesc& t, const TexLevelTaps& k, float out[4]) {
    if (FLOAT_TEXELS && ((t.info >> 5) & RDR_TEXD_FLOAT)) {
        // an SFLOAT image (Image.h:9-12): one float4 per texel, filtered as it is (no decode, no scale)
        const RDR_GLOBAL float* fx = as_global((const float*)t.base);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (!(MASK & (1 << c))) continue;
            const float c00 = fx[4u * k.o00 + c], c10 = fx[4u * k.o10 + c], c01 = fx[4u * k.o01 + c], c11 = fx[4u * k.o11 + c];
            const float top = fmaf(k.a, c10 - c00, c00), bot = fmaf(k.a, c11 - c01, c01);
            out[c] = fmaf(k.b, bot - top, top);
        }
        return;
    }
    const RDR_GLOBAL uint32_t* px = as_global((const uint32_t*)t.base);
    // the two texels of a row are neighbours in memory unless the footprint wraps around the edge
    uint32_t q00, q10, q01, q11;
    load_pair(px + k.o00, q00, q10);
    load_pair(px + k.o01, q01, q11);
    if (k.o10 != k.o00 + 1u) { q10 = px[k.o10]; q11 = px[k.o11]; }
    tex_filter<MASK>(srgb, ((t.info >> 5) & RDR_TEXD_SRGB) != 0u, q00, q10, q01, q11, k.a, k.b, out);
}
template <int MASK, bool FLOAT_TEXELS = false>
RDR_DEV void tex_fetch(const float* srgb, const DTexDesc& t, const TexTaps& k, float out[4]) {
    tex_fetch_level<MASK, FLOAT_TEXELS>(srgb, t, k.lv[0], out);
    if (k.two) {
        float c1[4];
        tex_fetch_level<MASK, FLOAT_TEXELS>(srgb, t, k.lv[1], c1);
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (MASK & (1 << c)) out[c] = fmaf(k.f, c1[c] - out[c], out[c]);
    }
}
// The same fetch in two steps, so that the loads of several textures are in flight together: tex_issue only loads
// the raw texels (the taps must have neighbouring pairs: tex_pairs_adjacent), tex_resolve filters them.
struct TexRaw { uint32_t q[2][4]; };
RDR_DEV bool tex_pairs_adjacent(const TexTaps& k) {
    return (int)(k.lv[0].o10 == k.lv[0].o00 + 1u) & ((int)!k.two | (int)(k.lv[1].o10 == k.lv[1].o00 + 1u));
}
RDR_DEV void tex_issue(const DTexDesc& t, const TexTaps& k, TexRaw& r) {
    const RDR_GLOBAL uint32_t* px = as_global((const uint32_t*)t.base);
    load_pair(px + k.lv[0].o00, r.q[0][0], r.q[0][1]);
    load_pair(px + k.lv[0].o01, r.q[0][2], r.q[0][3]);
    if (k.two) {
        load_pair(px + k.lv[1].o00, r.q[1][0], r.q[1][1]);
        load_pair(px + k.lv[1].o01, r.q[1][2], r.q[1][3]);
    }
}
template <int MASK>
RDR_DEV void tex_resolve(const float* srgb, const DTexDesc& t, const TexTaps& k, const TexRaw& r, float out[4]) {
    const bool is_srgb = ((t.info >> 5) & RDR_TEXD_SRGB) != 0u;
    tex_filter<MASK>(srgb, is_srgb, r.q[0][0], r.q[0][1], r.q[0][2], r.q[0][3], k.lv[0].a, k.lv[0].b, out);
    if (k.two) {
        float c1[4];
        tex_filter<MASK>(srgb, is_srgb, r.q[1][0], r.q[1][1], r.q[1][2], r.q[1][3], k.lv[1].a, k.lv[1].b, c1);
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (MASK & (1 << c)) out[c] = fmaf(k.f, c1[c] - out[c], out[c]);
    }
}
