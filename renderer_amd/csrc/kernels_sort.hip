// Device-wide exclusive scan and stable LSD radix sort of (key, value) pairs.
//
// Used by the binning stage: the (tile, triangle) pairs are produced in
// triangle submission order and sorted by tile with a STABLE sort, so every
// tile list keeps the reference's draw order (Vulkan rasterisation order),
// which the depth test LESS and the blend pass depend on.
//
// Sort layout: the input is cut into chunks of 64 lanes x SORT_ROUNDS keys; one
// 64-thread workgroup (a single wavefront) owns a chunk, so ranks inside a
// chunk come from wave ballots with no cross-wave ordering to maintain.
#include "device_common.h"
#include "kernels.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)
#define SCAN_RAW_MAX_TILES 2048u  // up to here every scan block sums the tile totals before it itself (8 M elements)
#define SORT_ROUNDS 16
#define SORT_CHUNK (64 * SORT_ROUNDS)
#define SORT_MAX_BINS 512

namespace {

RDR_DEV uint32_t wave_inclusive_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan of the block's 256 values; returns the block total in `total`.
RDR_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds_wave_sums, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) lds_wave_sums[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; w++) {
        const uint32_t s = lds_wave_sums[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_reduce(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t ws[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + k * SCAN_BLOCK + threadIdx.x;
        if (i < n) s += in[i];
    }
    uint32_t total;
    block_exclusive_scan(s, ws, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan of up to 1024*16 block sums in place, total to out_total[0].
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_sums(uint32_t* __restrict__ sums, uint32_t nb, uint32_t* __restrict__ out_total) {
    __shared__ uint32_t ws[SCAN_BLOCK / 64];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nb; base += SCAN_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, ws, total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) out_total[0] = carry;
}

// RAW: block_sums holds the tile totals as k_scan_reduce left them, and every block adds up the totals before its own
// itself (block 0 also writes the grand total). With a few hundred tiles that is cheaper than a single-block kernel in
// between, whose launch and drain cost more than its work.
template <bool RAW>
// `in` and `out` may be the same array (both callers scan in place): neither is declared restrict.
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply(const uint32_t* in, uint32_t n, const uint32_t* __restrict__ block_sums,
                                                           uint32_t* out, uint32_t* __restrict__ out_total) {
    __shared__ uint32_t ws[SCAN_BLOCK / 64];
    const uint32_t base = blockIdx.x * SCAN_TILE;
    uint32_t before = 0;
    if (RAW) {
        const uint32_t upto = blockIdx.x == 0 ? gridDim.x : blockIdx.x;  // block 0 has nothing before it: it sums everything
        uint32_t part = 0;
        for (uint32_t i = threadIdx.x; i < upto; i += SCAN_BLOCK) part += block_sums[i];
        uint32_t sum;
        block_exclusive_scan(part, ws, sum);
        if (blockIdx.x == 0) { if (threadIdx.x == 0) out_total[0] = sum; }
        else before = sum;
    } else before = block_sums[blockIdx.x];
    // thread t owns SCAN_ITEMS consecutive elements so that the running sum stays in registers
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
        v[k] = i < n ? in[i] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = before + block_exclusive_scan(s, ws, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const uint32_t i = base + threadIdx.x * SCAN_ITEMS + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}

// ---- radix sort -----------------------------------------------------------

__global__ void __launch_bounds__(64) k_sort_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t bins,
                                                  uint32_t* __restrict__ ghist, uint32_t nchunks) {
    __shared__ uint32_t hist[SORT_MAX_BINS];
    const uint32_t lane = threadIdx.x;
    for (uint32_t b = lane; b < bins; b += 64) hist[b] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * SORT_CHUNK;
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t i = base + r * 64 + lane;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & (bins - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t b = lane; b < bins; b += 64) ghist[(size_t)b * nchunks + blockIdx.x] = hist[b];
}

__global__ void __launch_bounds__(64) k_sort_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n,
                                                     uint32_t shift, uint32_t bits, const uint32_t* __restrict__ ghist_scanned,
                                                     uint32_t nchunks, uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals) {
    __shared__ uint32_t offs[SORT_MAX_BINS];
    const uint32_t lane = threadIdx.x;
    const uint32_t bins = 1u << bits;
    for (uint32_t b = lane; b < bins; b += 64) offs[b] = ghist_scanned[(size_t)b * nchunks + blockIdx.x];
    __syncthreads();
    const uint32_t base = blockIdx.x * SORT_CHUNK;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // the rounds depend on each other through offs[], so their loads are issued up front: one memory latency for
    // the chunk instead of one per round
    uint32_t kreg[SORT_ROUNDS], vreg[SORT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t i = base + r * 64 + lane;
        kreg[r] = i < n ? keys[i] : 0u;
        vreg[r] = i < n ? vals[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t i = base + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t key = kreg[r], val = vreg[r];
        const uint32_t digit = (key >> shift) & (bins - 1u);
        // lanes holding the same digit as this lane
        unsigned long long same = __ballot(valid);
        for (uint32_t b = 0; b < bits; b++) {
            const unsigned long long bal = __ballot(valid && ((digit >> b) & 1u));
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(same & lt_mask);
        uint32_t pos = 0;
        if (valid) pos = offs[digit] + rank;
        __syncthreads();  // every lane has read its bin before any leader advances it
        if (valid && rank == 0) offs[digit] += (uint32_t)__popcll(same);
        __syncthreads();
        if (valid) { out_keys[pos] = key; out_vals[pos] = val; }
    }
}

}  // namespace

size_t scan_scratch_words(uint32_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1; }

// out[i] = sum(in[0..i)), total -> d_total[0]. `scratch` holds scan_scratch_words(n) words. in may alias out.
void launch_exclusive_scan(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* scratch, uint32_t* d_total) {
    if (n == 0) { (void)hipMemsetAsync(d_total, 0, 4, s); return; }
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, scratch);
    if (nb <= SCAN_RAW_MAX_TILES) {
        hipLaunchKernelGGL(k_scan_apply<true>, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, (const uint32_t*)scratch, out, d_total);
    } else {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, scratch, nb, d_total);
        hipLaunchKernelGGL(k_scan_apply<false>, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, (const uint32_t*)scratch, out, d_total);
    }
}

uint32_t sort_chunks(uint32_t n) { return (n + SORT_CHUNK - 1) / SORT_CHUNK; }
size_t sort_hist_words(uint32_t n) { return (size_t)SORT_MAX_BINS * sort_chunks(n); }

// Stable sort of n pairs by the low `key_bits` bits of the key. keys/vals are
// ping-pong buffers [2]; returns the index of the buffer holding the result.
int launch_radix_sort(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], uint32_t n, uint32_t key_bits, uint32_t* hist,
                      uint32_t* scan_scratch, uint32_t* d_dummy_total) {
    if (n == 0) return 0;
    const uint32_t passes = (key_bits + 8) / 9;  // at most 9 bits (512 bins) per pass
    const uint32_t bits = (key_bits + passes - 1) / passes;
    const uint32_t nchunks = sort_chunks(n);
    int cur = 0;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * bits, bins = 1u << bits;
        hipLaunchKernelGGL(k_sort_hist, dim3(nchunks), dim3(64), 0, s, (const uint32_t*)keys[cur], n, shift, bins, hist, nchunks);
        launch_exclusive_scan(s, hist, hist, bins * nchunks, scan_scratch, d_dummy_total);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nchunks), dim3(64), 0, s, (const uint32_t*)keys[cur], (const uint32_t*)vals[cur], n, shift, bits,
                           (const uint32_t*)hist, nchunks, keys[cur ^ 1], vals[cur ^ 1]);
        cur ^= 1;
    }
    return cur;
}
