// oa_tri_fine.hpp -- the surface search once the pose has SETTLED: four lanes per query, one cell, whole triangles (round 6).
//
// Why: k_tri_search_grid (oa_tri.hpp) is built around its wave -- range lists, the shared scan, the pool and its flushes in 40 KB
// of LDS and 128 registers, four waves per SIMD -- and a wave's life is a chain of a dozen exposed memory round trips whatever
// its 64 queries need: with the pose settled (every query within a few per cent of an edge of its seed) a search still took
// 155 us + 37 us for the tree's leftovers, and neither the cell size nor the number of records moved that
// (profiles/r06a_surface_cell_sizes.txt).  A settled query needs almost nothing: its seed's distance d bounds the reach
// s = delta + sqrt(d^2 (1 + 1e-5)) (tri_reach_bound), and only triangles within s of the query can beat or tie the seed.
//
// The structure: a FINE uniform grid (cell edge h ~ half a mean bounding-box diagonal), kept sparse -- a hash table of the
// occupied cells only, {cell id, first record, records} per slot, open addressing -- whose cell lists are INFLATED: a triangle is listed
// in every cell within rho (Chebyshev) of its bounding box.  Then a query with s <= rho needs exactly ONE list, its own
// cell's: a triangle within s of p has a point x with |x - p|_inf <= s <= rho, x lies in the triangle's box and p in its
// cell, so the box inflated by rho meets the cell.  The records are WHOLE triangles (three float4: nine coordinates and the
// index): one hash probe, then one contiguous run of ~12 x 48 bytes read by the query's four lanes together, each record
// evaluated exactly -- no discs, no survivors, no pool, no second gather; no LDS and half the registers, so the chip holds
// twice the waves.  The answer is the lexicographic minimum of (d2, index) over a set that provably contains every
// triangle that can beat or tie: bit for bit what the general search and brute force return.
//
// What it does not settle -- no seed, s > rho (the pose still moves, or an outlier), a query outside the grid's box, a crowded
// cell (fans of thin triangles: more than `cap` records) -- goes on a list that k_tri_search_grid works through (`qlist`; the
// list itself: oa_tri.hpp, ulist_*), and from there to the triangle tree as before.  While the pose still moves by more
// than `gate` per iteration (DevState's translation / rotation rings) the launch returns at once and flags the list "everybody".
#pragma once
#include "oa_tri.hpp"

namespace oa {

struct FineParams {
    double lo[3];            // origin of the cell frame: the mesh's bounding box, lowered by rho (1 + 1e-6)
    double hi[3];            // ... and its far corner, raised likewise: queries outside [lo, hi] are not this search's
    double h, inv_h;         // cell edge
    double rho;              // every triangle is listed in all cells within rho of its bounding box
    double rho_query;        // a query is eligible when its reach is <= this (= rho (1 - 1e-6): the binning's own roundings)
    double scale, slack;     // as GridParams: the float evaluation's error bound delta = 64 u (scale + |p|_1) + slack
    double gate;             // the launch leaves everything to the general search while the pose moves by more than this per iteration
    int n[3];                // cells per axis (<= 1024: a cell id has 30 bits)
    int cap;                 // lists longer than this are left to the general search
    unsigned slots_mask;     // hash table: slots - 1 (a power of two)
    int hash_shift;          // 32 - log2(slots)
};

constexpr unsigned TFINE_EMPTY = 0xFFFFFFFFu;
constexpr int TFINE_MAX_PROBES = 256;

__host__ __device__ inline unsigned tfine_hash(unsigned id, int shift) { return (id * 0x9E3779B1u) >> shift; }

#if defined(__HIPCC__)

__device__ __forceinline__ void tfine_tri_range(const float4 *__restrict__ tri9, int t, const FineParams &fp, int lo[3], int hi[3], bool &ok)
{
    float a[3], b[3], c[3];
    load_tri(tri9, t, a, b, c);
    ok = true;
    for (int i = 0; i < 3; ++i) {
        const double mn = fmin(fmin((double)a[i], (double)b[i]), (double)c[i]) - fp.rho;
        const double mx = fmax(fmax((double)a[i], (double)b[i]), (double)c[i]) + fp.rho;
        if (!(mn <= mx)) ok = false;                                 // NaN vertex: the triangle can never be selected
        lo[i] = grid_cell_coord(mn, fp.lo[i], fp.inv_h, fp.n[i]);
        hi[i] = grid_cell_coord(mx, fp.lo[i], fp.inv_h, fp.n[i]);
    }
}

__device__ __forceinline__ unsigned tfine_cell_id(const FineParams &fp, int x, int y, int z) { return ((unsigned)z * (unsigned)fp.n[1] + (unsigned)y) * (unsigned)fp.n[0] + (unsigned)x; }

#if !defined(OA_FAMILY_TU) && defined(OA_EXPERIMENTS)
// pass 0: how many list entries the mesh makes (spread totals, as k_tri_grid_bin), and the most any one triangle makes
__global__ void k_tfine_total(const float4 *__restrict__ tri9, int n_tris, FineParams fp, unsigned long long *__restrict__ total)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n = 0;
    if (t < n_tris) {
        int lo[3], hi[3];
        bool ok;
        tfine_tri_range(tri9, t, fp, lo, hi, ok);
        if (ok) n = (unsigned long long)(hi[0] - lo[0] + 1) * (unsigned long long)(hi[1] - lo[1] + 1) * (unsigned long long)(hi[2] - lo[2] + 1);
    }
    __shared__ unsigned long long part[16];
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += part[w];
        if (sum) atomicAdd(total + (size_t)(blockIdx.x % TOTAL_SLOTS) * TOTAL_STRIDE, sum);
    }
}

// the slot of cell `id`: found or claimed (INSERT), or -1 (not there / the table is too crowded: `overflow` is raised)
template <bool INSERT>
__device__ __forceinline__ int tfine_slot(uint4 *__restrict__ table, const FineParams &fp, unsigned id, int *overflow)
{
    unsigned s = tfine_hash(id, fp.hash_shift);
    for (int probe = 0; probe < TFINE_MAX_PROBES; ++probe) {
        unsigned k = __hip_atomic_load(&table[s].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == id) return (int)s;
        if (k == TFINE_EMPTY) {
            if (!INSERT) return -1;
            k = atomicCAS(&table[s].x, TFINE_EMPTY, id);
            if (k == TFINE_EMPTY || k == id) return (int)s;
        }
        s = (s + 1u) & fp.slots_mask;
    }
    if (overflow) *overflow = 1;
    return -1;
}

// pass 1: the occupied cells claim their slots, counts[slot] = entries of the cell
__global__ void k_tfine_count(const float4 *__restrict__ tri9, int n_tris, FineParams fp, uint4 *__restrict__ table, int *__restrict__ counts,
                              int *__restrict__ overflow)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    int lo[3], hi[3];
    bool ok;
    tfine_tri_range(tri9, t, fp, lo, hi, ok);
    if (!ok) return;
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const int s = tfine_slot<true>(table, fp, tfine_cell_id(fp, x, y, z), overflow);
                if (s >= 0) atomicAdd(&counts[s], 1);
            }
}

// pass 2: the records.  counts[] runs down to zero (the list's length stays in offsets[slot + 1] - offsets[slot])
__global__ void k_tfine_fill(const float4 *__restrict__ tri9, int n_tris, FineParams fp, uint4 *__restrict__ table, int *__restrict__ counts,
                             const long long *__restrict__ offsets, float4 *__restrict__ rec)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    int lo[3], hi[3];
    bool ok;
    tfine_tri_range(tri9, t, fp, lo, hi, ok);
    if (!ok) return;
    const float4 u = tri9[3ll * t], v = tri9[3ll * t + 1], w = tri9[3ll * t + 2];
    const float4 w2 = make_float4(w.x, __uint_as_float((uint32_t)t), 0.f, 0.f);
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const int s = tfine_slot<false>(table, fp, tfine_cell_id(fp, x, y, z), nullptr);
                if (s < 0) continue;                                 // (only after an overflow: the host throws the build away)
                const long long pos = offsets[s] + (long long)(atomicSub(&counts[s], 1) - 1);
                rec[3 * pos] = u; rec[3 * pos + 1] = v; rec[3 * pos + 2] = w2;
            }
}

// pass 3: the slots learn where their lists start and how long they are
__global__ void k_tfine_finish(uint4 *__restrict__ table, int n_slots, const long long *__restrict__ offsets)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    if (table[s].x == TFINE_EMPTY) return;
    const long long first = offsets[s];
    table[s].y = (unsigned)first;
    table[s].z = (unsigned)(offsets[s + 1] - first);
}
#endif  // !OA_FAMILY_TU

// ---- the search -----------------------------------------------------------------------------------------------------------------
// FOUR lanes per query.  The four read their list together -- lane q takes float4 number 4 m + q of the next twelve (four records,
// 192 contiguous bytes) -- because a wave's load of 64 unrelated addresses costs the texture unit 64 tag look-ups whatever it
// returns (one lane per query: 400 us for a million settled queries, three times that many line fetches from the L2s as
// bytes used); then a transposition inside the quad (one select and one DPP move per word and slot) hands lane q record q, every lane
// evaluates its triangle exactly, and the quad's lanes merge (d2, index) at the end.  The seed, the reach and the hash probe are
// done by all four lanes alike (same addresses: one request).
// keys[i] = (bits(d2) << 32) | triangle for what is settled here (the searches' own format); the others go on the list above.
// stats (instrumented build): [0] settled, [1] no seed, [2] reach > rho, [3] outside the box, [4] cell not listed / over the cap,
// [5] records evaluated
#define OA_TFINE_QP(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, true))

template <bool STATS>
__global__ __launch_bounds__(256, 8) void k_tri_settle(const DevState *__restrict__ st, const float4 *__restrict__ src4, int ns, FineParams fp,
                                                       const uint4 *__restrict__ table, const float4 *__restrict__ rec,
                                                       const float4 *__restrict__ tri9, const int *__restrict__ prev,
                                                       unsigned long long *__restrict__ keys, int *__restrict__ ulist, int ulist_cap_,
                                                       int *__restrict__ counters, int *__restrict__ counters_next,
                                                       unsigned long long *__restrict__ stats)
{
    if (st->halt) return;
    if (blockIdx.x == 0 && threadIdx.x < ULIST_PARTS) counters_next[threadIdx.x * ULIST_STRIDE] = 0;   // the next launch's list starts empty
    {   // the gate: last iteration's step (translation + rotation x object size, in the target's local frame) -- the measure the
        // general search takes its record budget from
        const int last = (st->n + 4) % 5;
        const double moved = st->use_target && st->n > 0 ? (st->ring_t[last] + st->ring_r[last] * fp.scale) * st->local_per_world : 0.0;
        if (moved > fp.gate) {
            if (blockIdx.x == 0 && threadIdx.x == 0) counters[0] = -1;
            return;
        }
    }
    const int vb = xcd_block_index();
    const int gt = vb * (int)blockDim.x + threadIdx.x;
    const int i = gt >> 2, q = threadIdx.x & 3;
    const bool alive = i < ns;
    float pf[3] = { 0.f, 0.f, 0.f };
    int s = -1;
    if (alive) {
        const float4 p4 = src4[i];
        co_find(st, p4.x, p4.y, p4.z, pf[0], pf[1], pf[2]);
        s = prev[i];
    }
    float best = INFINITY;
    uint32_t bidx = IDX_NONE;
    int why = 1;                                                     // (STATS) why not: 1 no seed, 2 reach, 3 outside, 4 cell
    bool eligible = false;
    unsigned first = 0;
    int len = 0;
    if (s >= 0) {
        float a[3], b[3], c[3], r[3];
        load_tri(tri9, s, a, b, c);
        closest_on_tri(pf, a, b, c, r);
        const float d = tri_dist2(pf, r);
        if (d < INFINITY) {
            best = d; bidx = (uint32_t)s;
            // the reach, as k_tri_search_grid takes it: lim = the best so far or the search radius, whichever is smaller
            const double pabs = fabs((double)pf[0]) + fabs((double)pf[1]) + fabs((double)pf[2]);
            const double delta = 64.0 * 5.9604644775390625e-08 * (fp.scale + pabs) + fp.slack;
            const float lim = fminf(best, search_cutoff2(st, pf[0], pf[1], pf[2]));
            const double reach = tri_reach_bound(lim, delta);
            why = 2;
            if (reach <= fp.rho_query) {
                why = 3;
                const double px = pf[0], py = pf[1], pz = pf[2];
                if (px >= fp.lo[0] && px <= fp.hi[0] && py >= fp.lo[1] && py <= fp.hi[1] && pz >= fp.lo[2] && pz <= fp.hi[2]) {
                    why = 4;
                    const unsigned id = tfine_cell_id(fp, grid_cell_coord(px, fp.lo[0], fp.inv_h, fp.n[0]), grid_cell_coord(py, fp.lo[1], fp.inv_h, fp.n[1]),
                                                      grid_cell_coord(pz, fp.lo[2], fp.inv_h, fp.n[2]));
                    unsigned h = tfine_hash(id, fp.hash_shift);
                    for (int probe = 0; probe < TFINE_MAX_PROBES; ++probe) {
                        const uint4 e = table[h];
                        if (e.x == id) { eligible = (int)e.z <= fp.cap && e.z > 0u; first = e.y; len = (int)e.z; break; }
                        if (e.x == TFINE_EMPTY) break;
                        h = (h + 1u) & fp.slots_mask;
                    }
                }
            }
        }
    }
    if (!eligible) { first = 0; len = 0; }
    // The list, four records per step: float4 number f of the step is record f / 3's part f % 3; lane q loads numbers q, 4 + q, 8 + q
    // (clamped to the list's last float4: a lane past the end re-reads it and its record is not counted).  No branch around the
    // loads, and the next step's are in flight during this step's evaluations.
    const long long f0 = 3ll * first, f_last = f0 + (len > 0 ? 3ll * len - 1 : 0);
    auto ld = [&](long long f) { return rec[f < f_last ? f : f_last]; };
    float4 F0 = ld(f0 + q), F1 = ld(f0 + 4 + q), F2 = ld(f0 + 8 + q);
    int n_eval = 0;
    for (int j = 0; __any(j < len); j += 4) {
        const long long fn = f0 + 3ll * (j + 4);
        const float4 N0 = ld(fn + q), N1 = ld(fn + 4 + q), N2 = ld(fn + 8 + q);
        // transposition: requester q's part U sits in (lane, register) = (0,F0) (3,F0) (2,F1) (1,F2); V: (1,F0) (0,F1) (3,F1) (2,F2);
        // W: (2,F0) (1,F1) (0,F2) (3,F2).  Every source lane serves exactly one requester per part: it picks the register, one quad
        // permutation delivers.
        float4 U, V, W;
#define OA_TFINE_PICK(dst, r0, r1, r2, r3, ctrl)                                                                             \
        {                                                                                                                    \
            const float4 pick = q == 0 ? (r0) : (q == 1 ? (r1) : (q == 2 ? (r2) : (r3)));                                     \
            dst.x = OA_TFINE_QP(pick.x, ctrl); dst.y = OA_TFINE_QP(pick.y, ctrl); dst.z = OA_TFINE_QP(pick.z, ctrl); dst.w = OA_TFINE_QP(pick.w, ctrl); \
        }
        OA_TFINE_PICK(U, F0, F2, F1, F0, 0x6C)                        // quad_perm(0,3,2,1): lanes 0..3 send F0, F2, F1, F0
        OA_TFINE_PICK(V, F1, F0, F2, F1, 0xB1)                        // quad_perm(1,0,3,2)
        OA_TFINE_PICK(W, F2, F1, F0, F2, 0xC6)                        // quad_perm(2,1,0,3)
#undef OA_TFINE_PICK
        const float a[3] = { U.x, U.y, U.z }, b[3] = { U.w, V.x, V.y }, c[3] = { V.z, V.w, W.x };
        float r[3];
        closest_on_tri(pf, a, b, c, r);
        const float d = tri_dist2(pf, r);
        const uint32_t t = __float_as_uint(W.y);
        if (j + q < len) {
            if (STATS) ++n_eval;
            if (d < best || (d == best && t < bidx && d < INFINITY)) { best = d; bidx = t; }
        }
        F0 = N0; F1 = N1; F2 = N2;
    }
    // the quad's lanes agree: nearest triangle, lowest index on ties
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    const bool mine = alive && q == 0;
    if (mine && eligible) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    ulist_append(ulist, counters, ulist_cap_, vb * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6), mine && !eligible, i);
    if (STATS && stats) {
        int sv[6] = { mine && eligible, mine && !eligible && why == 1, mine && !eligible && why == 2, mine && !eligible && why == 3,
                      mine && !eligible && why == 4, n_eval };
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            for (int o = 32; o > 0; o >>= 1) sv[k] += __shfl_xor(sv[k], o, 64);
            if ((threadIdx.x & 63) == 0 && sv[k]) atomicAdd(stats + k, (unsigned long long)sv[k]);
        }
    }
}

#endif  // __HIPCC__
}  // namespace oa
