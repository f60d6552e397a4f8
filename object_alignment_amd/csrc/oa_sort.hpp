// oa_sort.hpp -- the stable ascending ORDER of up to 30-bit keys (an LSD radix argsort): the Morton sorts of the index builds
// (source slots; the tree's primitives), which sort (key, index) pairs and only ever read the sorted indices.
//
// Why not rocprim::radix_sort_pairs: on this part it costs 140-165 us for 1M pairs whichever algorithm it picks (merge sort
// up to 1Mi items: 21 launches; Onesweep forced: 139 us -- its decoupled look-back chains workgroups through memory across
// eight XCDs), 54 us for 100k (tools/sort_bench.hip) -- a fifth of a whole 1M <-> 1M alignment call for two sorts
// (tools/trace_calls.sh).  The data are 8 MB per pass: microseconds at HBM rates.
//
// Here: three passes of 10 bits, each three plain launches with no workgroup waiting for another:
//   k_sort_hist     per tile (256 threads x SORT_ITEMS items) the digit histogram -> hist[bin][tile]
//   k_sort_scan     one workgroup per bin: exclusive scan over the tiles, in place; totals[bin]
//   k_sort_scatter  per tile: bin bases (scan of the 1024 totals, redundantly per workgroup: 4 KB from L2), then every item's
//                   rank among the tile's items of the same digit IN ITEM ORDER (wave w owns a contiguous quarter of the tile;
//                   round by round the lanes with equal digits find each other with ten ballots, the lowest takes the wave's
//                   running count from LDS), then the waves' counts are chained in wave order and the items go out.
// STABLE, hence the permutation is the one rocprim's stable sorts return -- bit for bit the same slot order, primitive order
// and everything downstream (tools/sort_bench.hip compares them).  No atomics on global memory, no float arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oa {

constexpr int SORT_BITS = 10, SORT_BINS = 1 << SORT_BITS, SORT_THREADS = 256, SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_ITEMS = 8, SORT_TILE = SORT_THREADS * SORT_ITEMS, SORT_WAVE_ITEMS = 64 * SORT_ITEMS;

#if defined(__HIPCC__)

// the key of item i: passes 2.. read the packed {key, value} pairs the pass before wrote
template <bool PACKED>
__device__ __forceinline__ uint32_t sort_key_at(const void *__restrict__ in, long long i)
{
    return PACKED ? ((const uint2 *)in)[i].x : ((const uint32_t *)in)[i];
}

template <bool PACKED>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(const void *__restrict__ in, int n, int shift, int n_tiles, int *__restrict__ hist)
{
    __shared__ int h[SORT_BINS];
    for (int b = threadIdx.x; b < SORT_BINS; b += SORT_THREADS) h[b] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * SORT_TILE;
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long i = base + r * SORT_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(sort_key_at<PACKED>(in, i) >> shift) & (SORT_BINS - 1)], 1);   // (integer counts: the order of the atomics is immaterial)
    }
    __syncthreads();
    for (int b = threadIdx.x; b < SORT_BINS; b += SORT_THREADS) hist[(size_t)b * n_tiles + blockIdx.x] = h[b];
}

// inclusive scan over the workgroup's SORT_THREADS values (wave shuffles, then the waves' totals); `total` = their sum
__device__ __forceinline__ int sort_block_scan(int v, int *wave_tot /* __shared__ [SORT_WAVES] */, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    int before = 0;
    total = 0;
#pragma unroll
    for (int k = 0; k < SORT_WAVES; ++k) { const int t = wave_tot[k]; if (k < w) before += t; total += t; }
    __syncthreads();                                                 // wave_tot is reused by the caller's next call
    return v + before;
}

#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scan(int *__restrict__ hist, int n_tiles, int *__restrict__ totals)
{
    __shared__ int wave_tot[SORT_WAVES];
    int *row = hist + (size_t)blockIdx.x * n_tiles;
    int carry = 0;
    for (int base = 0; base < n_tiles; base += SORT_THREADS) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? row[i] : 0;
        int total;
        const int incl = sort_block_scan(v, wave_tot, total);
        if (i < n_tiles) row[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}
#endif  // !OA_FAMILY_TU

// PACKED_IN: the items are {key, value} pairs (else keys, the value of item i is i); LAST: only the values go out (the
// callers want the order, nobody reads the sorted keys), else {key, value} pairs -- ONE scattered store per item either way:
// a pass is bound by its scattered stores (every one its own cache line), two arrays cost twice.
template <bool PACKED_IN, bool LAST>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(const void *__restrict__ in, void *__restrict__ out, int n, int shift,
                                                               int n_tiles, const int *__restrict__ hist, const int *__restrict__ totals)
{
    __shared__ int cnt[SORT_WAVES][SORT_BINS];                      // a wave's running digit counts, later its first output position per digit
    __shared__ int wave_tot[SORT_WAVES];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tile = blockIdx.x;
    // the tile's items, wave w's quarter in rounds of 64 consecutive items: all loads in flight before anything else
    const long long first = (long long)tile * SORT_TILE + (long long)w * SORT_WAVE_ITEMS + lane;
    uint32_t key[SORT_ITEMS];
    int val[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const long long i = first + r * 64;
        key[r] = 0u; val[r] = (int)i;
        if (i < n) {
            if (PACKED_IN) { const uint2 kv = ((const uint2 *)in)[i]; key[r] = kv.x; val[r] = (int)kv.y; }
            else key[r] = ((const uint32_t *)in)[i];
        }
    }
    // where each digit's items start in the output: exclusive scan of the totals (thread t: bins 4t .. 4t+3) + this tile's offset
    constexpr int BPT = SORT_BINS / SORT_THREADS;
    int tb[BPT], run = 0;
#pragma unroll
    for (int k = 0; k < BPT; ++k) { tb[k] = totals[threadIdx.x * BPT + k]; run += tb[k]; }
    for (int b = threadIdx.x; b < SORT_WAVES * SORT_BINS; b += SORT_THREADS) (&cnt[0][0])[b] = 0;
    int total;
    const int incl = sort_block_scan(run, wave_tot, total);        // (its barriers also publish the zeroed counts)
    int bin_start[BPT];
    {
        int s = incl - run;
#pragma unroll
        for (int k = 0; k < BPT; ++k) { bin_start[k] = s + hist[(size_t)(threadIdx.x * BPT + k) * n_tiles + tile]; s += tb[k]; }
    }
    // ranks inside the wave's quarter, in item order
    int rank[SORT_ITEMS];
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        const bool valid = first + r * 64 < n;
        const int d = (int)((key[r] >> shift) & (SORT_BINS - 1));
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int k = 0; k < SORT_BITS; ++k) {
            const bool bit = (d >> k) & 1;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        rank[r] = 0;
        if (valid) {                                                 // (peers holds this lane: never empty; its lowest lane is valid too)
            const int leader = __ffsll((long long)peers) - 1;
            int old = 0;
            if (lane == leader) { old = cnt[w][d]; cnt[w][d] = old + __popcll(peers); }
            rank[r] = __shfl(old, leader, 64) + __popcll(peers & ((1ull << lane) - 1ull));
        }
    }
    __syncthreads();
    // the waves' counts chained in wave order behind the tile's start: cnt[w][d] = where wave w's first item of digit d goes
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        const int b = threadIdx.x * BPT + k;
        int s = bin_start[k];
#pragma unroll
        for (int ww = 0; ww < SORT_WAVES; ++ww) { const int c = cnt[ww][b]; cnt[ww][b] = s; s += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ITEMS; ++r) {
        if (first + r * 64 < n) {
            const int d = (int)((key[r] >> shift) & (SORT_BINS - 1));
            const int dest = cnt[w][d] + rank[r];
            if (LAST) ((int *)out)[dest] = val[r];
            else ((uint2 *)out)[dest] = make_uint2(key[r], (uint32_t)val[r]);
        }
    }
}

#endif  // __HIPCC__

// temporary storage of sort_order_lsd, in bytes: two buffers of {key, value} pairs and the histograms
inline size_t sort_order_tmp_bytes(size_t n)
{
    const size_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    return 2 * 8 * n + sizeof(int) * ((size_t)SORT_BINS * n_tiles + SORT_BINS) + 256;
}

#if defined(__HIPCC__)
// order[0 .. n) = the STABLE ascending order of keys[0 .. n) on their low `bits` bits (bits <= 30, the bits above them zero):
// order[j] = index of the j-th smallest key, equal keys by index -- what a stable sort_pairs returns for the values 0, 1, 2, ...
// The keys are left as they are.  tmp: sort_order_tmp_bytes(n) bytes, 8-byte aligned.  Everything on `stream`, no wait.
inline hipError_t sort_order_lsd(void *tmp, const uint32_t *keys, int *order, size_t n, int bits, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const int n_tiles = (int)((n + SORT_TILE - 1) / SORT_TILE);
    uint2 *buf[2] = { (uint2 *)tmp, (uint2 *)tmp + n };
    int *hist = (int *)(buf[1] + n);
    int *totals = hist + (size_t)SORT_BINS * n_tiles;
    const int passes = (bits + SORT_BITS - 1) / SORT_BITS;
    const dim3 grid((unsigned)n_tiles), blk(SORT_THREADS);
    const void *src = keys;
    for (int p = 0; p < passes; ++p) {
        const bool first = p == 0, last = p == passes - 1;
        void *dst = last ? (void *)order : (void *)buf[p & 1];
        const int shift = p * SORT_BITS;
        if (first) hipLaunchKernelGGL(k_sort_hist<false>, grid, blk, 0, stream, src, (int)n, shift, n_tiles, hist);
        else hipLaunchKernelGGL(k_sort_hist<true>, grid, blk, 0, stream, src, (int)n, shift, n_tiles, hist);
        hipLaunchKernelGGL(k_sort_scan, dim3(SORT_BINS), blk, 0, stream, hist, n_tiles, totals);
        if (first && last) hipLaunchKernelGGL((k_sort_scatter<false, true>), grid, blk, 0, stream, src, dst, (int)n, shift, n_tiles, (const int *)hist, (const int *)totals);
        else if (first) hipLaunchKernelGGL((k_sort_scatter<false, false>), grid, blk, 0, stream, src, dst, (int)n, shift, n_tiles, (const int *)hist, (const int *)totals);
        else if (last) hipLaunchKernelGGL((k_sort_scatter<true, true>), grid, blk, 0, stream, src, dst, (int)n, shift, n_tiles, (const int *)hist, (const int *)totals);
        else hipLaunchKernelGGL((k_sort_scatter<true, false>), grid, blk, 0, stream, src, dst, (int)n, shift, n_tiles, (const int *)hist, (const int *)totals);
        src = dst;
    }
    return hipGetLastError();
}
#endif

// ---- small inputs: the same stable order from ONE workgroup -----------------------------------------------------------------
// Up to SORT_SMALL_MAX keys: a bitonic sort of (key << 32 | index) in LDS -- the composite keys are distinct, so their
// ascending order IS the stable order of the keys.  One launch (~10 us) where the three-pass argsort above needs nine
// (29-40 us whatever n: tools/sort_bench.hip) -- an alignment call on a 2562-vertex mesh sorts twice in ~0.4 ms.
constexpr int SORT_SMALL_MAX = 8192, SORT_SMALL_THREADS = 1024;

#if defined(__HIPCC__)
#if !defined(OA_FAMILY_TU)
__global__ __launch_bounds__(SORT_SMALL_THREADS) void k_sort_small(const uint32_t *__restrict__ keys, int n, int *__restrict__ order)
{
    __shared__ unsigned long long a[SORT_SMALL_MAX];
    int P = 64;
    while (P < n) P <<= 1;                                           // padded to a power of two with keys that sort last
    for (int i = threadIdx.x; i < P; i += SORT_SMALL_THREADS)
        a[i] = i < n ? (((unsigned long long)keys[i] << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += SORT_SMALL_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), x = i | j;      // the pair (i, i ^ j), i < x
                const unsigned long long u = a[i], v = a[x];
                const bool up = (i & k) == 0;
                if ((u > v) == up) { a[i] = v; a[x] = u; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += SORT_SMALL_THREADS) order[i] = (int)(uint32_t)a[i];
}

// out[j] = src[idx[j]]
__global__ void k_gather_int(const int *__restrict__ src, const int *__restrict__ idx, int n, int *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = src[idx[j]];
}
#endif  // !OA_FAMILY_TU

// the stable ascending order of keys[0 .. n) on their low `bits` bits, whatever n: one workgroup for small inputs, the
// three-launch passes above otherwise.  tmp: sort_order_tmp_bytes(n) bytes (unused by the small path).
inline hipError_t sort_order(void *tmp, const uint32_t *keys, int *order, size_t n, int bits, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (n <= (size_t)SORT_SMALL_MAX) {
        hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(SORT_SMALL_THREADS), 0, stream, keys, (int)n, order);
        return hipGetLastError();
    }
    return sort_order_lsd(tmp, keys, order, n, bits, stream);
}
#endif

// ---- exclusive prefix sums of int counts into 64-bit offsets (the grids' cell starts) -------------------------------------------
// offsets[0 .. n] for counts[0 .. n): three plain launches, no workgroup waits for another (as the sort's passes):
//   k_scan_tile_sums   per tile of SCAN_TILE counts its sum
//   k_scan_tile_bases  one workgroup: exclusive scan of the tile sums, in place
//   k_scan_apply       per tile: its counts' exclusive scan + the tile's base; the last tile also writes offsets[n]
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
inline size_t scan_tmp_bytes(size_t n) { return sizeof(long long) * ((n + SCAN_TILE - 1) / SCAN_TILE + 1); }

#if defined(__HIPCC__)
#if !defined(OA_FAMILY_TU)
__device__ __forceinline__ long long scan_block_ll(long long v, long long *wave_tot, long long &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    long long before = 0;
    total = 0;
    for (int k = 0; k < nw; ++k) { const long long t = wave_tot[k]; if (k < w) before += t; total += t; }
    __syncthreads();
    return v + before;                                               // inclusive
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(const int *__restrict__ counts, long long n, long long *__restrict__ tile_sums)
{
    __shared__ long long wave_tot[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    long long s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) if (base + r < n) s += counts[base + r];
    long long total;
    (void)scan_block_ll(s, wave_tot, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_tile_bases(long long *__restrict__ tile_sums, long long n_tiles)
{
    __shared__ long long wave_tot[16];
    long long carry = 0;
    for (long long base = 0; base < n_tiles; base += 1024) {
        const long long i = base + threadIdx.x;
        const long long v = i < n_tiles ? tile_sums[i] : 0;
        long long total;
        const long long incl = scan_block_ll(v, wave_tot, total);
        if (i < n_tiles) tile_sums[i] = carry + incl - v;
        carry += total;
    }
    if (threadIdx.x == 0) tile_sums[n_tiles] = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const int *__restrict__ counts, long long n, const long long *__restrict__ tile_sums,
                                                             long long *__restrict__ offsets)
{
    __shared__ long long wave_tot[SCAN_THREADS / 64];
    const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    long long s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) { v[r] = base + r < n ? counts[base + r] : 0; s += v[r]; }
    long long total;
    long long run = scan_block_ll(s, wave_tot, total) - s + tile_sums[blockIdx.x];
#pragma unroll
    for (int r = 0; r < SCAN_ITEMS; ++r) { if (base + r < n) offsets[base + r] = run; run += v[r]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) offsets[n] = tile_sums[gridDim.x];
}
#endif  // !OA_FAMILY_TU

// offsets[0 .. n] = exclusive prefix sums of counts[0 .. n) (offsets[n] = the total); tmp: scan_tmp_bytes(n) bytes, 8-byte aligned
inline hipError_t scan_counts_ll(void *tmp, const int *counts, size_t n, long long *offsets, hipStream_t stream)
{
    long long *tile_sums = (long long *)tmp;
    const long long n_tiles = (long long)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (n_tiles == 0) return hipMemsetAsync(offsets, 0, sizeof(long long), stream);
    hipLaunchKernelGGL(k_scan_tile_sums, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, stream, counts, (long long)n, tile_sums);
    hipLaunchKernelGGL(k_scan_tile_bases, dim3(1), dim3(1024), 0, stream, tile_sums, n_tiles);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, stream, counts, (long long)n, (const long long *)tile_sums, offsets);
    return hipGetLastError();
}
#endif

}  // namespace oa
