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

// The scan in ONE launch: a block scans its tile, publishes the tile's sum, learns the sum of everything before the
// tile by looking back over what the blocks before it published (64 tiles per look, one per lane of its first wave),
// publishes its own running total and writes its outputs. status[tile] = state << 32 | value (0: nothing yet, 1: the
// tile's sum, 2: the sum up to and including the tile). The words must be zero when the kernel starts, and are zero
// again when it ends: every block counts itself out in status[tiles] once it has finished looking back, and the last one
// to do so clears the lot. Blocks are dispatched in index order and wait only for smaller indices; the wait is bounded
// (the sums are then wrong: it does not happen in a working launch).
#define SCAN_SPIN_LIMIT (1u << 20)
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_chained(const uint32_t* in, uint32_t n, uint32_t* out, unsigned long long* status,
                                                            uint32_t* __restrict__ out_total) {
    __shared__ uint32_t ws[SCAN_BLOCK / 64];
    __shared__ uint32_t lds_before;
    const uint32_t tile = blockIdx.x, tiles = gridDim.x, base = tile * SCAN_TILE;
    const uint32_t lane = threadIdx.x & 63u;
    // The tile is SCAN_ITEMS / 4 rows of SCAN_BLOCK four-element groups; thread t owns group t of every row, so that a
    // wave reads and writes 1 KiB runs (a thread owning sixteen consecutive elements made every load touch 64 lines).
    constexpr int ROWS = SCAN_ITEMS / 4;
    const bool whole = base + SCAN_TILE <= n && ((uintptr_t)in & 15u) == 0u && ((uintptr_t)out & 15u) == 0u;
    uint32_t v[ROWS][4];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const uint32_t i = base + ((uint32_t)r * SCAN_BLOCK + threadIdx.x) * 4u;
        if (whole) {
            const rdr_uint4_t q = *(const rdr_uint4_t*)(in + i);
            v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[r][k] = i + k < n ? in[i + k] : 0u;
        }
    }
    // exclusive position of every group inside the tile: a block scan per row, the rows one after the other
    uint32_t within[ROWS];
    uint32_t total = 0u;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        uint32_t row_total;
        within[r] = total + block_exclusive_scan((v[r][0] + v[r][1]) + (v[r][2] + v[r][3]), ws, row_total);
        total += row_total;
    }
    if (threadIdx.x < 64u) {  // the first wave looks back
        if (lane == 0u)
            __hip_atomic_store(status + tile, ((unsigned long long)(tile ? 1u : 2u) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t before = 0u, c = tile, spins = 0u;  // tiles [0, c) still to be accounted for
        while (c) {
            const bool mine = lane < c;  // lane j looks at tile c - 1 - j
            const unsigned long long w = mine ? __hip_atomic_load(status + (c - 1u - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            const uint32_t state = (uint32_t)(w >> 32);
            const unsigned long long totals = __ballot(mine && state == 2u), missing = __ballot(mine && state == 0u);
            // usable: the lanes up to the nearest running total (all of the window if there is none)
            const unsigned long long upto = totals ? ((totals & (0ull - totals)) << 1) - 1ull : ~0ull;
            if (missing & upto) {
                if (++spins > SCAN_SPIN_LIMIT) break;
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            uint32_t part = (mine && ((upto >> lane) & 1ull)) ? (uint32_t)w : 0u;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d);
            before += part;
            if (totals) break;
            c -= min(c, 64u);
        }
        if (lane == 0u) {
            if (tile) __hip_atomic_store(status + tile, (2ull << 32) | (unsigned long long)(before + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_before = before;
            if (tile + 1u == tiles) out_total[0] = before + total;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const uint32_t i = base + ((uint32_t)r * SCAN_BLOCK + threadIdx.x) * 4u;
        const uint32_t e0 = lds_before + within[r], e1 = e0 + v[r][0], e2 = e1 + v[r][1], e3 = e2 + v[r][2];
        if (whole) {
            rdr_uint4_t q;
            q.x = e0; q.y = e1; q.z = e2; q.w = e3;
            *(rdr_uint4_t*)(out + i) = q;
        } else {
            if (i < n) out[i] = e0;
            if (i + 1u < n) out[i + 1u] = e1;
            if (i + 2u < n) out[i + 2u] = e2;
            if (i + 3u < n) out[i + 3u] = e3;
        }
    }
    // count out; the last block leaves the status words as it found them
    __shared__ uint32_t lds_last;
    if (threadIdx.x == 0u) {
        __threadfence();
        lds_last = atomicAdd((unsigned int*)(status + tiles), 1u) + 1u == tiles ? 1u : 0u;
    }
    __syncthreads();
    if (lds_last)
        for (uint32_t i = threadIdx.x; i <= tiles; i += SCAN_BLOCK) status[i] = 0ull;
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


// ---- radix sort in one launch per digit ----------------------------------------------------------------------------
// The pair sort of the binning stage is a handful of small dependent kernels, and what it costs is their number. Here a
// digit pass is ONE kernel: a 512-thread block owns OS_CHUNK consecutive pairs, counts its digits, and gets the number
// of pairs of every digit in the chunks before it by looking back over the counts those blocks publish (a chained
// scan: a block first publishes its own counts, then walks towards chunk 0 adding up counts until it meets a block
// that already knows its running total, then publishes its own running total). One kernel in front of the passes
// (k_os_hist) counts every digit of every pass once, which gives each pass the first position of each digit.
//
// A status word is epoch << 32 | state << 30 | count: one 64-bit store publishes it, and a word left by an earlier
// sort is recognised by its epoch, so the status array is never cleared. Blocks are dispatched in index order and
// only wait for blocks with smaller indices; the wait is bounded all the same (OS_SPIN_LIMIT reads of one word), and a
// block that gives up raises *error and stores nothing outside the output arrays.
#define OS_THREADS 512
#define OS_WAVES (OS_THREADS / 64)
#define OS_ROUNDS 8
#define OS_CHUNK (OS_THREADS * OS_ROUNDS)  // 4096 pairs
#define OS_MAX_PASSES 4
#define OS_AGGREGATE 1ull
#define OS_PREFIX 2ull
#define OS_SPIN_LIMIT (1u << 20)
#define OS_WINDOW 8  // status words of that many chunks fetched together when looking back

// Digit counts of all passes (ghist[pass * SORT_MAX_BINS + digit], zero on entry) and, while the keys go by, the tile
// ranges cleared for k_ranges (`clear` words from `to_clear` on).
__global__ void __launch_bounds__(OS_THREADS) k_os_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t passes, uint32_t bits,
                                                       uint32_t* __restrict__ ghist, uint32_t* __restrict__ to_clear, uint32_t clear) {
    __shared__ uint32_t hist[OS_MAX_PASSES * SORT_MAX_BINS];
    for (uint32_t i = threadIdx.x; i < passes * SORT_MAX_BINS; i += OS_THREADS) hist[i] = 0u;
    for (uint32_t i = blockIdx.x * OS_THREADS + threadIdx.x; i < clear; i += gridDim.x * OS_THREADS) to_clear[i] = 0u;
    __syncthreads();
    const uint32_t mask = (1u << bits) - 1u, lane = threadIdx.x & 63u, base = blockIdx.x * OS_CHUNK;
    uint32_t k[OS_ROUNDS];
#pragma unroll
    for (int r = 0; r < OS_ROUNDS; r++) {
        const uint32_t i = base + r * OS_THREADS + threadIdx.x;
        k[r] = i < n ? keys[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < OS_ROUNDS; r++) {
        const bool valid = base + r * OS_THREADS + threadIdx.x < n;  // a prefix of the lanes
        for (uint32_t p = 0; p < passes; p++) {
            // Neighbouring pairs are tiles of one triangle: their upper digits are mostly equal, and 64 additions to
            // one LDS word take 64 turns. The first lane of every run of equal digits adds the run's length instead.
            const uint32_t digit = (k[r] >> (p * bits)) & mask;
            const uint32_t left = (uint32_t)__shfl_up((int)digit, 1);
            const unsigned long long heads = __ballot(valid && (lane == 0u || digit != left));
            const uint32_t n_valid = (uint32_t)__popcll(__ballot(valid));
            if (valid && ((heads >> lane) & 1ull)) {
                const unsigned long long later = lane == 63u ? 0ull : heads >> (lane + 1u);  // heads of the runs behind this one
                const uint32_t len = later ? (uint32_t)__ffsll((long long)later) : n_valid - lane;
                atomicAdd(&hist[p * SORT_MAX_BINS + digit], len);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < passes * SORT_MAX_BINS; i += OS_THREADS)
        if (hist[i]) atomicAdd(&ghist[i], hist[i]);
}

// One digit pass. ghist: this pass's digit counts over all keys. status: [chunk][SORT_MAX_BINS] words, see above.
__global__ void __launch_bounds__(OS_THREADS) k_os_pass(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t n, uint32_t shift,
                                                       uint32_t bits, const uint32_t* __restrict__ ghist, unsigned long long* status, uint32_t epoch,
                                                       uint32_t* __restrict__ out_keys, uint32_t* __restrict__ out_vals, uint32_t* error) {
    __shared__ uint32_t offs[OS_WAVES][SORT_MAX_BINS];  // per wave: running count of a digit, later its first position
    __shared__ uint32_t ws[OS_WAVES];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t bins = 1u << bits;
    for (uint32_t i = tid; i < OS_WAVES * SORT_MAX_BINS; i += OS_THREADS) (&offs[0][0])[i] = 0u;
    // wave w owns pairs [base + w * 64 * OS_ROUNDS, + 64 * OS_ROUNDS) of the chunk, in rounds of 64: pair order is (wave, round, lane)
    const uint32_t base = blockIdx.x * OS_CHUNK + wave * (64u * OS_ROUNDS);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t kreg[OS_ROUNDS], vreg[OS_ROUNDS], rank[OS_ROUNDS];
#pragma unroll
    for (int r = 0; r < OS_ROUNDS; r++) {
        const uint32_t i = base + r * 64 + lane;
        kreg[r] = i < n ? keys[i] : 0u;
        vreg[r] = i < n ? vals[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OS_ROUNDS; r++) {
        const bool valid = base + r * 64 + lane < n;
        const uint32_t digit = (kreg[r] >> shift) & (bins - 1u);
        unsigned long long same = __ballot(valid);  // lanes holding the same digit as this lane
        for (uint32_t b = 0; b < bits; b++) {
            const unsigned long long bal = __ballot(valid && ((digit >> b) & 1u));
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const uint32_t before = (uint32_t)__popcll(same & lt_mask);
        rank[r] = valid ? offs[wave][digit] + before : 0u;
        // A wave's row of counts is its own, and LDS takes a wave's instructions in order: waiting for the reads to
        // return keeps them in front of the leaders' additions, no block-wide barrier needed.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (valid && before == 0u) offs[wave][digit] += (uint32_t)__popcll(same);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // thread d looks after digit d: the waves' counts become the waves' first positions inside the chunk's run of d
    uint32_t count = 0u;
    if (tid < bins) {
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) { const uint32_t c = offs[w][tid]; offs[w][tid] = count; count += c; }
    }
    const unsigned long long tag = (unsigned long long)epoch << 32;
    unsigned long long* const mine = status + (size_t)blockIdx.x * SORT_MAX_BINS + tid;
    if (tid < bins) __hip_atomic_store(mine, tag | ((blockIdx.x ? OS_AGGREGATE : OS_PREFIX) << 30) | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // first position of digit d in the output: the keys with a smaller digit (exclusive scan of the totals over the block)
    const uint32_t total = tid < bins ? ghist[tid] : 0u;
    uint32_t first;
    {
        const uint32_t inc = wave_inclusive_scan(total, (int)lane);
        if (lane == 63u) ws[wave] = inc;
        __syncthreads();
        uint32_t before = 0u;
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) if ((uint32_t)w < wave) before += ws[w];
        first = before + inc - total;
    }
    // pairs of digit d in the chunks before this one
    uint32_t earlier = 0u;
    if (tid < bins && blockIdx.x) {
        // OS_WINDOW chunks at a time, nearest first: their words are fetched together (one trip to the cache), then taken
        // in order up to the first that is not there yet (fetched again, with what lies beyond it) or knows its total
        uint32_t spins = 0u, c = blockIdx.x;  // chunks [0, c) are still to be accounted for
        bool done = false;
        while (!done && c) {
            unsigned long long w[OS_WINDOW];
#pragma unroll
            for (int j = 0; j < OS_WINDOW; j++)
                w[j] = (uint32_t)j < c ? __hip_atomic_load(status + (size_t)(c - 1u - (uint32_t)j) * SORT_MAX_BINS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (int j = 0; j < OS_WINDOW; j++) {
                if (done || (uint32_t)j >= c) break;
                const uint32_t state = (uint32_t)(w[j] >> 30) & 3u;
                if ((w[j] >> 32) != epoch || !state) {  // not published yet
                    c -= (uint32_t)j;
                    if (++spins > OS_SPIN_LIMIT) { atomicOr(error, 1u); done = true; }
                    else __builtin_amdgcn_s_sleep(1);
                    goto again;
                }
                earlier += (uint32_t)(w[j] & 0x3FFFFFFFull);
                if (state == (uint32_t)OS_PREFIX) done = true;
            }
            if (!done) c -= min(c, (uint32_t)OS_WINDOW);
        again:;
        }
        __hip_atomic_store(mine, tag | (OS_PREFIX << 30) | (unsigned long long)(earlier + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < bins) {
#pragma unroll
        for (int w = 0; w < OS_WAVES; w++) offs[w][tid] += first + earlier;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < OS_ROUNDS; r++) {
        if (base + r * 64 + lane >= n) continue;
        const uint32_t pos = offs[wave][(kreg[r] >> shift) & (bins - 1u)] + rank[r];
        if (pos < n) { out_keys[pos] = kreg[r]; out_vals[pos] = vreg[r]; }  // pos >= n only after a failed wait
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

// One launch (k_scan_chained). `status`: scan_chained_words(n) words that are zero, and are zero again afterwards.
size_t scan_chained_words(uint32_t n) { return 2 * ((size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 1); }
void launch_exclusive_scan_chained(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* status, uint32_t* d_total) {
    if (n == 0) { (void)hipMemsetAsync(d_total, 0, 4, s); return; }
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_chained, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n, out, (unsigned long long*)status, d_total);
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

// The same sort with one launch per digit plus one in front (see k_os_pass). n < 2^30. `work` holds
// sort_onesweep_work_bytes(n) bytes, of which the status words must have been zero when the buffer was new (they are
// not cleared between sorts); *epoch is the caller's running pass number for this buffer (it goes up by one per digit;
// the caller clears the buffer and the counter long before 2^32). `to_clear` / `clear`: words
// zeroed on the way (the tile ranges k_ranges fills), saving the fill a launch. *error is set to non-zero if a block
// gave up waiting (it never happens in a working launch; the result is then undefined but in bounds).
uint32_t sort_onesweep_chunks(uint32_t n) { return (n + OS_CHUNK - 1) / OS_CHUNK; }
size_t sort_onesweep_work_bytes(uint32_t n) { return (size_t)OS_MAX_PASSES * SORT_MAX_BINS * 4 + (size_t)sort_onesweep_chunks(n) * SORT_MAX_BINS * 8; }
int launch_radix_sort_onesweep(hipStream_t s, uint32_t* keys[2], uint32_t* vals[2], uint32_t n, uint32_t key_bits, void* work, uint32_t* epoch,
                               uint32_t* to_clear, uint32_t clear, uint32_t* error, bool ghist_zeroed) {
    if (n == 0) return 0;
    const uint32_t passes = (key_bits + 8) / 9, bits = (key_bits + passes - 1) / passes, chunks = sort_onesweep_chunks(n);
    uint32_t* ghist = (uint32_t*)work;
    unsigned long long* status = (unsigned long long*)(ghist + OS_MAX_PASSES * SORT_MAX_BINS);
    if (!ghist_zeroed) (void)hipMemsetAsync(ghist, 0, (size_t)passes * SORT_MAX_BINS * 4, s);
    hipLaunchKernelGGL(k_os_hist, dim3(chunks), dim3(OS_THREADS), 0, s, (const uint32_t*)keys[0], n, passes, bits, ghist, to_clear, clear);
    int cur = 0;
    for (uint32_t p = 0; p < passes; p++) {
        ++*epoch;  // the caller clears the status words before the counter wraps
        hipLaunchKernelGGL(k_os_pass, dim3(chunks), dim3(OS_THREADS), 0, s, (const uint32_t*)keys[cur], (const uint32_t*)vals[cur], n, p * bits, bits,
                           (const uint32_t*)(ghist + p * SORT_MAX_BINS), status, *epoch, keys[cur ^ 1], vals[cur ^ 1], error);
        cur ^= 1;
    }
    return cur;
}
