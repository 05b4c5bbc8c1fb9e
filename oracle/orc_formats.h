// TEST INFRASTRUCTURE — CPU restatement ("oracle"), see orc_math.h header.
//
// Storage-format conversions that the Vulkan driver performs for the
// reference's attachment formats (Renderer/VulkanRenderer.cpp:556-652):
// RGBA16F colour, B10G11R11_UFLOAT bloom/env cubes, RGBA8 UNORM/SRGB textures.
// The driver is third-party code absent from /root/reference; these follow the
// Vulkan 1.3 specification ("Floating-Point Format Conversions", "sRGB
// transfer functions") with the rounding choices stated per function.
// parity unpinned: the reference holds no vectors for these conversions.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// float32 -> float16, round to nearest even, overflow to inf, NaN kept quiet.
inline uint16_t float_to_half(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to >= 65520 -> inf
    if (ax < 0x33000001u) return (uint16_t)sign;                // < 2^-25 (or == 2^-25 ties to even 0)
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;  // 24-bit significand
    int shift;                                   // bits to drop
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // subnormal half
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7FFFFFu; }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q));  // carry into exponent is correct by construction
}
inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f;  // m * 2^-24
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7F800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
inline float quantize_half(float f) { return half_to_float(float_to_half(f)); }

// float32 -> unsigned small float with 5 exponent bits and MB mantissa bits
// (MB = 6 for the 11-bit R/G channels, 5 for the 10-bit B channel).
// Rules: negative and -inf -> 0; NaN -> NaN pattern; +inf -> inf; finite
// values above the largest finite -> largest finite; otherwise the mantissa is
// TRUNCATED (round toward zero); values below 2^-14 become denormals.
template <int MB> inline uint32_t float_to_ufloat(float f) {
    uint32_t x = f2u(f);
    const uint32_t max_finite = (30u << MB) | ((1u << MB) - 1u);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (31u << MB) | 1u;  // NaN
    if (x & 0x80000000u) return 0;                                 // negative (incl. -0, -inf)
    if (x == 0x7F800000u) return 31u << MB;                        // +inf
    int e = (int)(x >> 23) - 127;
    uint32_t m = x & 0x7FFFFFu;
    if (e > 15) return max_finite;
    if (e >= -14) return ((uint32_t)(e + 15) << MB) | (m >> (23 - MB));
    // denormal: value / 2^(-14-MB), truncated
    int shift = (23 - MB) + (-14 - e);
    if (shift > 24) return 0;
    return (m | 0x800000u) >> shift;
}
template <int MB> inline float ufloat_to_float(uint32_t v) {
    uint32_t e = v >> MB, m = v & ((1u << MB) - 1u);
    if (e == 0) return (float)m * u2f((uint32_t)(127 - 14 - MB) << 23);
    if (e == 31) return u2f(0x7F800000u | (m << (23 - MB)));
    return u2f(((e + 112u) << 23) | (m << (23 - MB)));
}
// VK_FORMAT_B10G11R11_UFLOAT_PACK32: R in bits 0..10, G in 11..21, B in 22..31
inline uint32_t pack_r11g11b10(float r, float g, float b) {
    return float_to_ufloat<6>(r) | (float_to_ufloat<6>(g) << 11) | (float_to_ufloat<5>(b) << 22);
}
inline void unpack_r11g11b10(uint32_t p, float& r, float& g, float& b) {
    r = ufloat_to_float<6>(p & 0x7FFu);
    g = ufloat_to_float<6>((p >> 11) & 0x7FFu);
    b = ufloat_to_float<5>(p >> 22);
}

// UNORM8: clamp to [0,1], scale, round half up. NaN -> 0.
inline uint8_t float_to_unorm8(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 1.0f) return 255;
    return (uint8_t)(int)(f * 255.0f + 0.5f);
}

// sRGB EOTF, evaluated in double once per code value and rounded to float.
struct SrgbTables {
    float decode[256];     // code -> linear
    float threshold[256];  // threshold[n] = linear value at code n+0.5; encode(x) = #{n : x >= threshold[n]}
    SrgbTables() {
        for (int i = 0; i < 256; i++) {
            decode[i] = (float)eotf((double)i / 255.0);
            threshold[i] = (float)eotf(((double)i + 0.5) / 255.0);
        }
    }
    static double eotf(double c) { return c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4); }
    uint8_t encode(float x) const {
        if (!(x > 0.0f)) return 0;
        int lo = 0, hi = 255;  // smallest n in [0,255] with x < threshold[n]; 255 if none below
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (x < threshold[mid]) hi = mid; else lo = mid + 1;
        }
        return (uint8_t)lo;
    }
};

}  // namespace orc
