// Shared by the tile rasteriser (kernels_raster.hip) and the blended pass (kernels_shade.hip): a staged triangle
// record rewritten for one 16x16 tile.
//
// E(P) = dX (Py - Yi) - dY (Px - Xi) needs 64 bits in general, but inside one tile it is E(O) + A u + B v with O the
// tile's origin, A = -dY, B = dX and 0 <= u, v < 4096 + 224 sub-pixels. For an unclipped triangle whose edges all
// have |A| + |B| <= 2^18 (1024 px) the variable part stays below 2^30, so E(O) - bias is folded into a 32-bit
// constant C (saturated at +-2^30, where the sign no longer depends on u, v) and the per-sample test becomes
// C + A u + B v >= 0 in 32-bit multiply-adds, with the same sign as the 64-bit expression. Clipped polygons and
// very long edges keep the 64-bit path.
#pragma once
#include "device_common.h"
#include "rdr_device.h"

enum { RASTER_EXACT64 = 0u, RASTER_COVER = 1u, RASTER_LOCAL32 = 2u };

struct RasterLocal {
    int A[3], B[3], C[3];
    int off[3][4];   // A sx + B sy for the four standard sample positions
    uint32_t box;    // x0 | y0 << 4 | (width - 1) << 8 | (65536 / width + 1) << 12, tile-local
    uint32_t id;
    uint32_t box4;   // x0 | y0 << 4 | x1 << 8 | y1 << 12: the record's pixel box inside the tile, tile-local
    double dzdX, dzdY, z00;
    uint32_t pad2[2];
};
static_assert(sizeof(RasterLocal) == sizeof(DTriRec) && sizeof(DTriRec) == 128, "rewritten in place");

// `slot` is a 128-byte DTriRec in LDS whose pixel box overlaps the tile at (ox0, oy0). If the record qualifies, it is
// rewritten in place as a RasterLocal (the caller is the slot's only reader until the next barrier) and
// RASTER_LOCAL32 is returned; otherwise it is left alone and RASTER_EXACT64 is returned. S = samples per pixel (for
// S == 1 the pixel-centre offset is folded into C).
template <int S>
RDR_DEV uint32_t raster_localise(uint32_t* slot, int ox0, int oy0) {
    const DTriRec* r = (const DTriRec*)slot;
    if ((r->n_view & 255u) != 3u) return RASTER_EXACT64;
    const int sxo[4] = {96, 224, 32, 160}, syo[4] = {32, 96, 160, 224};
    RasterLocal f;
    const int Ox = 256 * ox0, Oy = 256 * oy0;
    bool narrow = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = i == 2 ? 0 : i + 1;
        const int Xi = r->X[i], Yi = r->Y[i];
        const int dX = r->X[j] - Xi, dY = r->Y[j] - Yi;
        narrow = narrow && (abs(dX) + abs(dY) <= (1 << 18));
        const long long bias = ((dY < 0) || (dY == 0 && dX > 0)) ? 0 : 1;
        long long c = (long long)dX * (long long)(Oy - Yi) - (long long)dY * (long long)(Ox - Xi) - bias;
        if (S == 1) c += ((long long)dX - (long long)dY) * 128;  // the one sample sits at the pixel centre
        c = c > (1ll << 30) ? (1ll << 30) : (c < -(1ll << 30) - 1 ? -(1ll << 30) - 1 : c);
        f.A[i] = -dY; f.B[i] = dX; f.C[i] = (int)c;
#pragma unroll
        for (int sm = 0; sm < 4; sm++) f.off[i][sm] = -dY * sxo[sm] + dX * syo[sm];
    }
    if (!narrow) return RASTER_EXACT64;
    const int bx0 = max((int)r->px0, ox0), bx1 = min((int)r->px1, ox0 + RDR_TILE - 1);
    const int by0 = max((int)r->py0, oy0), by1 = min((int)r->py1, oy0 + RDR_TILE - 1);
    const uint32_t bw = (uint32_t)(bx1 - bx0 + 1);
    f.box = (uint32_t)(bx0 - ox0) | ((uint32_t)(by0 - oy0) << 4) | ((bw - 1u) << 8) | ((65536u / bw + 1u) << 12);
    f.box4 = (uint32_t)(bx0 - ox0) | ((uint32_t)(by0 - oy0) << 4) | ((uint32_t)(bx1 - ox0) << 8) | ((uint32_t)(by1 - oy0) << 12);
    f.id = r->id;
    f.dzdX = r->dzdX; f.dzdY = r->dzdY; f.z00 = r->z00;
    f.pad2[0] = f.pad2[1] = 0u;
    *(RasterLocal*)slot = f;
    return RASTER_LOCAL32;
}
