// oa_tri.hpp -- closest point on the target's triangle SURFACE (closes discrepancy D2).
//
// This is what Blender's BVHTree.find_nearest (functions/general.py:297) returns: the nearest point of the nearest
// triangle of the evaluated base mesh.  The per-triangle arithmetic restates Blender's closest_on_tri_to_point_v3
// (Ericson, "Real-Time Collision Detection" 5.1.5) in float32 with explicit operation order and no fma -- identical,
// bit for bit, to the CPU oracle (oracle/oa_oracle.c: oo_closest_on_tri).  Nearest triangle wins, lowest triangle
// index on ties.  Blender API knowledge: PARITY UNPINNED against Blender itself (not installed, not vendored).
//
// Search: triangles are binned into every cell of a uniform grid their bounding box overlaps; a query scans rings of
// cells around its (box-projected) position exactly like k_nn_search_grid.  A triangle that has not been seen after
// ring r has a bounding box disjoint from the searched cube, so all of it is at real distance >= sqrt(|p-pc|^2 + m^2);
// the float32 evaluation can undershoot the real distance by at most delta = 64 u (|coords|), hence the stop rule
// (sqrt(|p-pc|^2 + m^2) - delta)^2 (1 - 1e-5) > lim  (lim = best so far, or the search radius of DevState::cut_a).
// Each cell-list entry carries the triangle's bounding sphere; candidates are filtered on those contiguous records
// and the survivors evaluated in a second phase (tri_queue_flush).  Queries the grid cannot settle -- far from the
// surface, or in crowded cells -- are finished by the triangle tree (oa_bvh.hpp); k_tri_search_all (every triangle
// for every query) is what OA_SEARCH_BRUTE runs.
#pragma once
#include "oa_grid.hpp"

namespace oa {

#if defined(__HIPCC__)

#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_pack_tris(const float *__restrict__ xyz, int n_verts, const int *__restrict__ tris, int n_tris,
                            float4 *__restrict__ tri9, int *__restrict__ bad)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    float q[9];
    for (int k = 0; k < 3; ++k) {
        const int v = tris[3ll * t + k];
        if (v < 0 || v >= n_verts) { atomicAdd(bad, 1); for (int j = 0; j < 3; ++j) q[3 * k + j] = NAN; continue; }
        for (int j = 0; j < 3; ++j) q[3 * k + j] = xyz[3ll * v + j];
    }
    tri9[3ll * t] = make_float4(q[0], q[1], q[2], q[3]);
    tri9[3ll * t + 1] = make_float4(q[4], q[5], q[6], q[7]);
    tri9[3ll * t + 2] = make_float4(q[8], 0.f, 0.f, 0.f);
}
#endif  // !OA_FAMILY_TU

// Totals of the mesh builds (sum of diagonals, number of cell-list entries) are accumulated in TOTAL_SLOTS words, 128 bytes
// apart, workgroup b adding to slot b mod TOTAL_SLOTS; the host adds the slots.  One word for everybody serialises: 7646
// workgroup atomics on it were ~90 us of a 1.96M-triangle launch (one per wave before that: 17 us already at 82k triangles).
constexpr int TOTAL_SLOTS = 64, TOTAL_STRIDE = 16;                  // (stride in 8-byte words)

// sum of triangle bounding-box diagonals (for the cell size) -- double atomics are fine here (one-time, not a result)
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_tri_diag_sum(const float4 *__restrict__ tri9, int n_tris, double *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double d = 0.0;
    if (t < n_tris) {
        float a[3], b[3], c[3];
        load_tri(tri9, t, a, b, c);
        double s = 0.0;
        for (int i = 0; i < 3; ++i) {
            const double lo = fmin(fmin((double)a[i], (double)b[i]), (double)c[i]);
            const double hi = fmax(fmax((double)a[i], (double)b[i]), (double)c[i]);
            s += (hi - lo) * (hi - lo);
        }
        d = sqrt(s);
        if (!(d < INFINITY)) d = 0.0;
    }
    // one atomic per WORKGROUP, spread over TOTAL_SLOTS words
    __shared__ double part[16];
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        if (t > 0.0) atomicAdd(out + (size_t)(blockIdx.x % TOTAL_SLOTS) * TOTAL_STRIDE, t);
    }
}
#endif  // !OA_FAMILY_TU

__device__ __forceinline__ void tri_cell_range(const float4 *__restrict__ tri9, int t, const GridParams &gp, int lo[3], int hi[3], bool &ok)
{
    float a[3], b[3], c[3];
    load_tri(tri9, t, a, b, c);
    ok = true;
    for (int i = 0; i < 3; ++i) {
        const double mn = fmin(fmin((double)a[i], (double)b[i]), (double)c[i]);
        const double mx = fmax(fmax((double)a[i], (double)b[i]), (double)c[i]);
        if (!(mn <= mx)) ok = false;                                 // NaN vertex: the triangle can never be selected
        lo[i] = grid_cell_coord(mn, gp.lo[i], gp.inv_h, gp.n[i]);
        hi[i] = grid_cell_coord(mx, gp.lo[i], gp.inv_h, gp.n[i]);
    }
}

// Cell-list record of a triangle: two float4 {cx, cy, cz, r} {nx, ny, nz, bits(triangle index)}.
//   (c, r)  smallest enclosing disc of the triangle in its own plane (centre rounded to float, radius measured from
//           the rounded centre and rounded UP): every point of the triangle is within r of c
//   n       unit normal scaled by (1 - 1e-6) (so |n| <= 1 after rounding), or 0 when the triangle is degenerate or its
//           corners deviate from the plane through c by more than eps_plane (then the record is a plain sphere)
// For x in the triangle:  |n . (p - x)| >= |n . (p - c)| - eps_plane   and   |perp(p - x)| >= |perp(p - c)| - r,
// which gives the lower bound tri_record_bound2() -- tight where the bounding sphere is not: for a query far (compared
// with the triangle size) from a smooth surface, dozens of triangles have spheres within reach but only the handful
// around the foot point pass this test.  NaN corners give NaN records: never skipped, never selected.
__device__ __forceinline__ void tri_record(const float *a, const float *b, const float *c, double eps_plane, uint32_t t,
                                           float4 &rec0, float4 &rec1)
{
    const double A[3] = { a[0], a[1], a[2] }, B[3] = { b[0], b[1], b[2] }, C[3] = { c[0], c[1], c[2] };
    double ab[3], ac[3], bc[3];
    for (int k = 0; k < 3; ++k) { ab[k] = B[k] - A[k]; ac[k] = C[k] - A[k]; bc[k] = C[k] - B[k]; }
    const double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2], ac2 = ac[0] * ac[0] + ac[1] * ac[1] + ac[2] * ac[2],
                 bc2 = bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2];
    const double N[3] = { ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0] };
    const double n2 = N[0] * N[0] + N[1] * N[1] + N[2] * N[2];
    double ctr[3];
    if (ab2 >= ac2 + bc2 || !(n2 > 0.0)) for (int k = 0; k < 3; ++k) ctr[k] = 0.5 * (A[k] + B[k]);       // right / obtuse at C, or degenerate
    else if (ac2 >= ab2 + bc2) for (int k = 0; k < 3; ++k) ctr[k] = 0.5 * (A[k] + C[k]);
    else if (bc2 >= ab2 + ac2) for (int k = 0; k < 3; ++k) ctr[k] = 0.5 * (B[k] + C[k]);
    else {                                                          // acute: circumcentre = A + (|ac|^2 (N x ab) + |ab|^2 (ac x N)) / (2 |N|^2)
        const double u[3] = { N[1] * ab[2] - N[2] * ab[1], N[2] * ab[0] - N[0] * ab[2], N[0] * ab[1] - N[1] * ab[0] };
        const double v[3] = { ac[1] * N[2] - ac[2] * N[1], ac[2] * N[0] - ac[0] * N[2], ac[0] * N[1] - ac[1] * N[0] };
        for (int k = 0; k < 3; ++k) ctr[k] = A[k] + (ac2 * u[k] + ab2 * v[k]) / (2.0 * n2);
    }
    if (!(n2 > 0.0)) {                                              // degenerate: the longest edge's midpoint
        if (ac2 >= ab2 && ac2 >= bc2) for (int k = 0; k < 3; ++k) ctr[k] = 0.5 * (A[k] + C[k]);
        else if (bc2 >= ab2 && bc2 >= ac2) for (int k = 0; k < 3; ++k) ctr[k] = 0.5 * (B[k] + C[k]);
    }
    const float cf[3] = { (float)ctr[0], (float)ctr[1], (float)ctr[2] };
    double r2 = 0.0;
    const double *V[3] = { A, B, C };
    for (int i = 0; i < 3; ++i) {
        double sdist = 0.0;
        for (int k = 0; k < 3; ++k) { const double d = V[i][k] - (double)cf[k]; sdist += d * d; }
        r2 = sdist > r2 ? sdist : r2;
    }
    if (!(r2 == r2) || A[0] != A[0] || B[0] != B[0] || C[0] != C[0]) r2 = NAN;   // fmax-style selects would hide a NaN corner
    const double r = sqrt(r2) * (1.0 + 1e-6) + 1e-37;
    float rf = r < 3.0e38 ? (float)r : INFINITY;                    // NaN stays NaN
    if ((double)rf < r) rf = nextafterf(rf, INFINITY);
    float nf[3] = { 0.f, 0.f, 0.f };
    if (n2 > 0.0 && n2 < 1e300) {
        const double inv = (1.0 - 1e-6) / sqrt(n2);
        const float cand[3] = { (float)(N[0] * inv), (float)(N[1] * inv), (float)(N[2] * inv) };
        double e = 0.0;
        for (int i = 0; i < 3; ++i) {
            double dd = 0.0;
            for (int k = 0; k < 3; ++k) dd += (double)cand[k] * (V[i][k] - (double)cf[k]);
            e = fabs(dd) > e ? fabs(dd) : e;
        }
        const double len2 = (double)cand[0] * cand[0] + (double)cand[1] * cand[1] + (double)cand[2] * cand[2];
        if (e <= eps_plane && len2 <= 1.0 && len2 >= 1.0 - 1e-5) { nf[0] = cand[0]; nf[1] = cand[1]; nf[2] = cand[2]; }
    }
    rec0 = make_float4(cf[0], cf[1], cf[2], rf);
    rec1 = make_float4(nf[0], nf[1], nf[2], __uint_as_float(t));
}

// The index word of a record also says whether the record's cell is the LOWEST cell of its triangle along x / y / z: a scan
// over a box of cells meets a triangle once per cell it shares with the box, and "lowest cell of the triangle or lowest cell
// of the box, along every axis" picks exactly one of those records without loading the triangle (oa_tri_ring.hpp).  The
// searches mask the flags off where they take the index (tri_candidate); build_tri_grid refuses meshes of 2^28 triangles.
constexpr uint32_t TRI_REC_INDEX_MASK = 0x0FFFFFFFu, TRI_REC_FLAG_X = 1u << 28, TRI_REC_FLAG_Y = 1u << 29, TRI_REC_FLAG_Z = 1u << 30;

// pass 0: counts[cell] += 1 for every cell overlapped; pass 1: write the triangle's record (disc, normal, index -- the
// search discards most candidates from these 32 contiguous bytes) at cell_start[cell] + cursor++
template <bool FILL>
__global__ void k_tri_grid_bin(const float4 *__restrict__ tri9, int n_tris, GridParams gp, int *__restrict__ counts,
                               const int *__restrict__ cell_start, float4 *__restrict__ cell_rec,
                               unsigned long long *__restrict__ total)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    int lo[3] = { 0, 0, 0 }, hi[3] = { -1, -1, -1 };               // (empty ranges: a thread without a triangle stays for the count's barrier)
    bool ok = false;
    if (t < n_tris) tri_cell_range(tri9, t, gp, lo, hi, ok);
    if (!ok) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = -1; }
    if (FILL && !ok) return;
    float4 rec0 = make_float4(0.f, 0.f, 0.f, 0.f), rec1 = rec0;
    if (FILL) {
        float a[3], b[3], c[3];
        load_tri(tri9, t, a, b, c);
        tri_record(a, b, c, (double)gp.eps_plane, (uint32_t)t, rec0, rec1);
    }
    unsigned long long n = 0;
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const int cidx = (z * gp.n[1] + y) * gp.n[0] + x;
                if (FILL) {
                    const long long pos = cell_start[cidx] + atomicAdd(&counts[cidx], 1);
                    cell_rec[2 * pos] = rec0;
                    float4 r1 = rec1;
                    r1.w = __uint_as_float((uint32_t)t | (x == lo[0] ? TRI_REC_FLAG_X : 0u) | (y == lo[1] ? TRI_REC_FLAG_Y : 0u) | (z == lo[2] ? TRI_REC_FLAG_Z : 0u));
                    cell_rec[2 * pos + 1] = r1;
                } else atomicAdd(&counts[cidx], 1);
                ++n;
            }
    if (!FILL && total) {
        // one atomic per WORKGROUP, spread over TOTAL_SLOTS words (the launch's critical path ends in a read-back of them)
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
}

__device__ __forceinline__ void tri_eval(const float *p, const float4 *__restrict__ tri9, uint32_t t, float &best, uint32_t &bidx)
{
    float a[3], b[3], c[3], r[3];
    load_tri(tri9, t, a, b, c);
    closest_on_tri(p, a, b, c, r);
    const float d = tri_dist2(p, r);
    if (d < best || (d == best && t < bidx && d < INFINITY)) { best = d; bidx = t; }
}

// How far from the query anything that can still beat or tie `lim` may lie, as a real distance: the float32 closest-point
// evaluation is >= (D - delta)^2 (1 - 1e-5) for real distance D, so whatever is farther than
//     s = delta + sqrt((lim + 1e-30) / (1 - 1e-5))
// is out.  An UPPER bound of s: the searches call this whenever a lane's best improves -- the whole wave walks through it --
// and in double (a division and a square root, three times over for the three thresholds derived from it) it was 190
// instructions, a fifth of k_tri_search_grid's VALU work.  Now: the quotient as a float product rounded up (1.0000102f covers
// 1 / (1 - 1e-5) and the product's rounding; 1.1e-30f the same for the absolute term), v_sqrt_f32 (1 ulp) with 3e-7 on top.
__device__ __forceinline__ double tri_reach_bound(float lim, double delta)
{
    const float x = lim * 1.0000102f + 1.1e-30f;
    return delta + (double)__builtin_amdgcn_sqrtf(x) * (1.0 + 3e-7);
}

// Squared-gap threshold above which a triangle's bounding box proves it can neither beat nor tie `best`: s^2; the extra
// 3e-6 covers the float rounding of the gap (6u) and of the threshold itself.
__device__ __forceinline__ float tri_skip_threshold(float best, double delta)
{
    if (!(best < INFINITY)) return INFINITY;
    const double s = tri_reach_bound(best, delta);
    return (float)(s * s * (1.0 + 3e-6));
}

struct TriSearchState {
    float best; uint32_t bidx;
    float lim, thr, reach;    // lim = min(best, search radius^2); thr = squared-gap threshold; reach >= sqrt(thr)
    float reach2f;            // s^2, rounded up: rows / rings whose squared gap exceeds it are out (the per-row arithmetic is float, GridQuery)
};

// everything derived from `lim` (called when the best improves)
__device__ __forceinline__ void tri_state_refresh(TriSearchState &s, double delta)
{
    if (s.lim < INFINITY) {
        const double r = tri_reach_bound(s.lim, delta);
        s.thr = (float)(r * r * (1.0 + 3e-6) * (1.0 + 1.1e-6));       // (x 1 / 0.999999: the records' bounds come unscaled)
        // sqrt(thr) <= r sqrt(1 + 3e-6) (1 + u) < r (1 + 1.6e-6); rounded to float and up: the sphere test's reach
        const double rr = r * (1.0 + 3e-6) + 1e-37;
        s.reach = rr < 3.0e38 ? (float)rr : INFINITY;
        const double r2 = r * r * (1.0 + 1.1e-6);
        s.reach2f = r2 < 3.0e38 ? (float)r2 : INFINITY;
    } else { s.thr = INFINITY; s.reach = INFINITY; s.reach2f = INFINITY; }
}

// Cell-list candidates go through two phases so that divergence does not multiply the expensive part.
//
// Phase 1 (tri_candidate): two tests on the contiguous 32-byte record -- the bounding sphere, then for its survivors the
// plane / in-plane-disc bound.  What is left goes into a POOL SHARED BY THE WAVE (LDS): (owner lane, triangle, bound).
//
// Phase 2 (tri_pool_flush), when the pool fills up and at the end of every batch of rows:
//   A  every lane evaluates its own most promising survivor (smallest bound) -- that alone usually brings its best
//      within a hair of the final answer;
//   B  the pool is re-tested against the owners' improved thresholds and compacted (LDS only);
//   C  the remaining contenders are evaluated 64 at a time by WHOEVER IS FREE: lane e takes pool entry e, fetches the
//      owner's query point with a cross-lane read, and merges (d2, triangle) into the owner's slot with a 64-bit LDS
//      atomic min -- lexicographic, i.e. nearest triangle, lowest index on ties, whatever the order.
// A wave therefore pays ceil(contenders of all its lanes / 64) closest-point evaluations instead of the maximum over
// its lanes per flush (PMC before: 65-80 evaluation trips per wave in the first iterations of a run, for 10-12
// evaluations per query).
// A pool entry is 6 bytes since round 6: (triangle << 6) | owner lane in one word (so the grid search takes meshes of up to 2^26
// triangles; larger ones go through the tree alone) and the bound's upper 16 bits -- rounded DOWN, so the re-test of phase 2 B is a
// little more permissive, never wrong.  Rounds 2-5: 9 bytes (index, owner, bound) and 288 entries in the same LDS.
constexpr int TRI_POOL = 432;      // pool entries per wave; a flush is due above TRI_POOL - 128
constexpr uint32_t TRI_POOL_MAX_TRIS = 1u << 26;
constexpr int TRI_SEGS = 10;       // cell-list ranges of one batch of rows (9 rows of the first block, or 5 rows x 2 end cells)

struct TriPool {                   // this wave's part of the workgroup's LDS
    unsigned *to;                  // (triangle << 6) | owner lane
    unsigned short *key;           // upper half of the bound's bits (a non-negative float, rounded down)
    unsigned long long *slot;      // one per lane: (bits(d2) << 32) | triangle of what others evaluated for it
    unsigned long long *first;     // one per lane: (bits(bound) << 32) | pool entry of its most promising survivor (~0: none)
    float4 *q4;                    // one per lane: its query and its reach (TriSearchState::reach), for whoever tests records on its behalf
    int n;                         // entries (wave-uniform)
};

// Lower bound of the squared distance from p to a triangle with record (rec0, rec1), given D2 = |p - c|^2 (float).
// Derivation in tri_record's comment; every rounding is covered: products and sums of floats are within 4u of their
// operands' magnitudes (u = 2^-24), the margins below are 1e-6 relative plus eps_plane + 8u rs absolute, where
// rs >= |p - c| (the caller passed the sphere test D2 <= rs^2 (1 + 3e-6)).
// (dx, dy, dz) = c - p as the sphere test formed it: only |n . (p - c)| and its square are used, and negation is exact.
__device__ __forceinline__ float tri_record_bound2(float dx, float dy, float dz, const float4 rec0, const float4 rec1, float D2, float rs,
                                                   float eps_plane)
{
    // (fma chains here and in the sphere test: these are bounds, not the oracle's arithmetic -- fewer roundings than the 4u the
    //  margins allow for, and two instructions less each)
    const float pdc = __builtin_fmaf(dz, rec1.z, __builtin_fmaf(dy, rec1.y, dx * rec1.x));   // -n . (p - c), |n| <= 1
    const float pd = fmaxf(fabsf(pdc) - (eps_plane + 4.8e-7f * rs), 0.f);         // >= 0, <= plane distance of every point
    // |perp(p - c)|^2 >= D2 - (n^ . (p - c))^2, and (n^ . v)^2 <= (n . v)^2 (1 + 5e-6) because |n| >= 1 - 2.5e-6
    // (the 2e-6 off D2: 4u for its own rounding, and 2 |pdc| e + e^2 <= 8e-7 D2 for the absolute error e <= 6u |p - c| of pdc)
    const float t2 = __builtin_fmaf(pdc * pdc, -1.000006f, D2 * 0.999998f);
    float lb = pd * pd;
    // the foot of p outside the disc: v_sqrt_f32 itself (1 ulp; the correctly rounded sequence around it is 19 instructions, per
    // record and for the whole wave).  t2 <= r^2 gives tg <= 0, t2 < 0 a NaN: neither adds anything.
    const float tg = __builtin_amdgcn_sqrtf(t2) * 0.9999996f - rec0.w;
    if (tg > 0.f) lb = __builtin_fmaf(tg, tg, lb);
    return lb;                                                                    // (its last 1e-6 is in TriSearchState::thr)
}

// phase 1 for one record per lane (valid = this lane has one), tested FOR lane `owner` of the wave: (px, py, pz) is the
// owner's query, reach / thr its thresholds (TriSearchState) -- the lane that runs the test need not be the owner
// (tri_scan_shared deals the wave's records out evenly).  Called by the whole wave.
__device__ __forceinline__ void tri_candidate(float px, float py, float pz, const float4 rec0, const float4 rec1, bool valid,
                                              float reach, float thr, int owner, float eps_plane, TriPool &pool, int *surv = nullptr)
{
    bool keep = false;
    float lb = 0.f;
    if (valid) {
        const float dx = rec0.x - px, dy = rec0.y - py, dz = rec0.z - pz;
        const float D2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const float rs = rec0.w + reach;
        if (!(D2 > rs * rs * 1.000003f)) {                         // else: farther than radius + reach, cannot beat or tie
            if (surv) *surv += 1 << 16;                            // (instrumented build: sphere passes in the high half)
            lb = tri_record_bound2(dx, dy, dz, rec0, rec1, D2, rs, eps_plane);
            keep = !(lb > thr);                                    // else: the plane / disc bound rules it out
        }
    }
    const unsigned long long m = __ballot(keep);
    if (keep) {
        if (surv) ++*surv;
        const int lane = threadIdx.x & 63;
        const int dst = pool.n + __popcll(m & ((1ull << lane) - 1ull));
        pool.to[dst] = ((__float_as_uint(rec1.w) & TRI_REC_INDEX_MASK) << 6) | (unsigned)owner;
        pool.key[dst] = (unsigned short)(__float_as_uint(lb) >> 16);
        // the owner's most promising survivor: smallest bound (lb >= +0: its bits order like its value; a NaN bound -- never
        // from a binned triangle -- sorts last and is evaluated with the rest)
        atomicMin(&pool.first[owner], ((unsigned long long)__float_as_uint(lb) << 32) | (unsigned long long)(unsigned)dst);
    }
    pool.n += __popcll(m);
}

// closest-point evaluation of triangle t (already loaded) against the running best
__device__ __forceinline__ void tri_consider(const float *p, const float4 u, const float4 v, const float4 w, uint32_t t,
                                             TriSearchState &s, double delta, float cutf, int *ev)
{
    if (t == s.bidx) return;                                       // (already the best: listed in several cells)
    const float a[3] = { u.x, u.y, u.z }, b[3] = { u.w, v.x, v.y }, c[3] = { v.z, v.w, w.x };
    float r[3];
    if (ev) ++*ev;
    closest_on_tri(p, a, b, c, r);
    const float d = tri_dist2(p, r);
    if (d < s.best || (d == s.best && t < s.bidx && d < INFINITY)) {
        const bool closer = d < s.best;
        s.best = d; s.bidx = t;
        if (closer) { s.lim = fminf(s.best, cutf); tri_state_refresh(s, delta); }
    }
}

// Phase 2 (see above).  Called by the whole wave.  (ev / trips: counters of the instrumented build)
__device__ __forceinline__ void tri_pool_flush(const float *p, const float4 *__restrict__ tri9, TriSearchState &s,
                                               TriPool &pool, double delta, float cutf, int *ev = nullptr, int *trips = nullptr)
{
    pool.n = __builtin_amdgcn_readfirstlane(pool.n);              // the same in every lane: the pool is the wave's
    if (pool.n == 0) return;
    const int lane = threadIdx.x & 63;
    // A) the most promising survivor of every lane
    const unsigned long long fs = __hip_atomic_load(&pool.first[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const int kslot = fs == ~0ull ? -1 : (int)(uint32_t)fs;
    if (kslot >= 0 && !(__uint_as_float((uint32_t)(fs >> 32)) > s.thr)) {
        const uint32_t t = pool.to[kslot] >> 6;
        const float4 u = tri9[3ll * t], v = tri9[3ll * t + 1], w = tri9[3ll * t + 2];
        tri_consider(p, u, v, w, t, s, delta, cutf, ev);
    }
    if (trips) ++*trips;
    // B) what is still a contender under its owner's improved best, compacted in place
    int n2 = 0;
    for (int base = 0; base < pool.n; base += 64) {
        const int e = base + lane;
        bool c = false;
        int own = 0, t = 0;
        float key = 0.f;
        unsigned to = 0u;
        unsigned short kb = 0;
        if (e < pool.n) { to = pool.to[e]; kb = pool.key[e]; own = (int)(to & 63u); t = (int)(to >> 6); key = __uint_as_float((unsigned)kb << 16); }
        const float othr = __shfl(s.thr, own, 64);
        const int oks = __shfl(kslot, own, 64), obi = __shfl((int)s.bidx, own, 64);
        if (e < pool.n) c = e != oks && t != obi && !(key > othr);
        const unsigned long long m = __ballot(c);
        if (c) {                                                   // dst <= e, and this trip's reads are done: in place is safe
            const int dst = n2 + __popcll(m & ((1ull << lane) - 1ull));
            pool.to[dst] = to; pool.key[dst] = kb;
        }
        n2 += __popcll(m);
    }
    // C) the contenders, 64 per trip, evaluated by whoever is free
    if (n2 > 0) {
        __hip_atomic_store(&pool.slot[lane], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        for (int base = 0; base < n2; base += 64) {
            const int e = base + lane;
            int own = 0;
            uint32_t t = 0;
            if (e < n2) { const unsigned to = pool.to[e]; own = (int)(to & 63u); t = to >> 6; }
            const float q0 = __shfl(p[0], own, 64), q1 = __shfl(p[1], own, 64), q2 = __shfl(p[2], own, 64);
            if (e < n2) {
                const float4 u = tri9[3ll * t], v = tri9[3ll * t + 1], w = tri9[3ll * t + 2];
                const float qo[3] = { q0, q1, q2 };
                const float a[3] = { u.x, u.y, u.z }, b[3] = { u.w, v.x, v.y }, c[3] = { v.z, v.w, w.x };
                float r[3];
                if (ev) ++*ev;
                closest_on_tri(qo, a, b, c, r);
                const float d = tri_dist2(qo, r);
                if (d < INFINITY)                                   // d >= +0: its bits order like the value; NaN / inf never win
                    atomicMin(&pool.slot[own], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)t);
            }
            if (trips) ++*trips;
        }
        const unsigned long long k = __hip_atomic_load(&pool.slot[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (k != ~0ull) {
            const float d = __uint_as_float((uint32_t)(k >> 32));
            const uint32_t t = (uint32_t)k;
            if (d < s.best || (d == s.best && t < s.bidx)) {
                const bool closer = d < s.best;
                s.best = d; s.bidx = t;
                if (closer) { s.lim = fminf(s.best, cutf); tri_state_refresh(s, delta); }
            }
        }
    }
    pool.n = 0;
    __hip_atomic_store(&pool.first[lane], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (pool.q4) __hip_atomic_store(&pool.q4[lane].w, s.reach, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // what the helpers test against from here on
}

// Phase 1 over the cell-list ranges the lanes have collected (per thread: seg_j[0 .. n_seg) first positions in cell_rec,
// seg_n[..] lengths).
// 26-bit positions: a chunk of the shared scan is (position << 6) | owner lane in one word.  (Records are 32 bytes and are
// addressed through 32-bit byte offsets: 2^27 would fit those.)  The host pads cell_rec with TRI_REC_PAD zeroed records --
// triangle 0 -- so that a chunk may always read four records.
constexpr long long TRI_REC_MAX_ENTRIES = (1ll << 26) - 16;
constexpr int TRI_REC_PAD = 4;
constexpr int TRI_SHARE_Q = 6;     // chunks (of four records) a lane hands to the wave per round of the shared scan
__device__ __forceinline__ float4 tri_ld_rec(const float4 *__restrict__ base, int entry, int half)
{
    return *(const float4 *)((const char *)base + ((unsigned)entry * 32u + (unsigned)half * 16u));
}

// every lane walks ITS ranges, four records per trip; the wave leaves when every lane is through: a wave pays
// max-over-lanes(records) trips (round 2 / 3; kept for A/B: OA_TRI_SHARE=0)
template <int BT>
__device__ __forceinline__ void tri_scan_segments(const float *p, const float4 *__restrict__ cell_rec,
                                                  const float4 *__restrict__ tri9, TriSearchState &s, float eps_plane,
                                                  const int (*seg_j)[BT], const unsigned short (*seg_n)[BT], int &n_seg, TriPool &pool,
                                                  double delta, float cutf, int *surv = nullptr, int *ev = nullptr, int *trips = nullptr)
{
    const int lane = threadIdx.x & 63;
    int k = 0, j = 0, end = 0;
    while (true) {
        if (j >= end && k < n_seg) { j = seg_j[k][threadIdx.x]; end = j + (int)seg_n[k][threadIdx.x]; ++k; }
        const bool active = j < end;
        if (!__any(active)) break;
        // four records (32 bytes each) per lane: eight independent loads; the clamped repeats of the last record are not tested
        const int last = active ? end - 1 : 0, jj = active ? j : 0;
        const int e1 = min(jj + 1, last), e2 = min(jj + 2, last), e3 = min(jj + 3, last);
        float4 a0, a1, b0, b1, c0, c1, d0, d1;
        a0 = a1 = b0 = b1 = c0 = c1 = d0 = d1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {                                              // (32-bit byte offsets from the uniform base: TRI_REC_MAX_ENTRIES)
            a0 = tri_ld_rec(cell_rec, jj, 0); a1 = tri_ld_rec(cell_rec, jj, 1); b0 = tri_ld_rec(cell_rec, e1, 0); b1 = tri_ld_rec(cell_rec, e1, 1);
            c0 = tri_ld_rec(cell_rec, e2, 0); c1 = tri_ld_rec(cell_rec, e2, 1); d0 = tri_ld_rec(cell_rec, e3, 0); d1 = tri_ld_rec(cell_rec, e3, 1);
        }
        tri_candidate(p[0], p[1], p[2], a0, a1, active, s.reach, s.thr, lane, eps_plane, pool, surv);
        tri_candidate(p[0], p[1], p[2], b0, b1, active && j + 1 < end, s.reach, s.thr, lane, eps_plane, pool, surv);
        if (pool.n > TRI_POOL - 128) tri_pool_flush(p, tri9, s, pool, delta, cutf, ev, trips);      // each test adds <= 64 entries
        tri_candidate(p[0], p[1], p[2], c0, c1, active && j + 2 < end, s.reach, s.thr, lane, eps_plane, pool, surv);
        tri_candidate(p[0], p[1], p[2], d0, d1, active && j + 3 < end, s.reach, s.thr, lane, eps_plane, pool, surv);
        if (pool.n > TRI_POOL - 128) tri_pool_flush(p, tri9, s, pool, delta, cutf, ev, trips);
        if (active) j += 4;
    }
    n_seg = 0;
}

// The same records, dealt out EVENLY and read COALESCED (round 4).
// (i) The lists of a wave's 64 queries differ in length -- settled, the longest is twice the mean (60 records against 28), in
// the first iterations 277 against 196 -- and the per-lane walk above keeps the whole wave for the longest.  (ii) In that
// walk every lane reads its own 128 bytes per trip: each of the eight load instructions of a trip touches 64 different cache
// lines, two tag look-ups per record, and the phase stamps of the instrumented build show the scan's time following the
// number of such look-ups, not the instructions (profiles/r04b_surface_phases.txt).
// Here the wave works through its records as one pool of work: in rounds, every lane cuts the next (up to) Q chunks of four
// consecutive records off its own ranges and writes them -- (position << 6) | owner lane -- into the wave's chunk table in LDS
// at its place in the wave-wide order (ballot prefix sums).  Then the table is walked 64 RECORDS per step: lane l takes record
// l mod 4 of chunk l / 4, so the four lanes of a quad read one 128-byte stretch and the quads of a range's consecutive chunks
// consecutive stretches (a load instruction touches ~16-24 lines instead of 64).  The lane fetches its chunk's owner's query
// and reach from the wave's owner table in LDS (TriPool::q4: one 16-byte read), tests its record, and a survivor goes into
// the wave's pool under its OWNER's name -- which is all phase 2 ever looked at.  Four steps per trip (four records per lane,
// from four different chunks).  A wave pays ceil(records of all its lanes / 256) trips instead of max-over-lanes(records) / 4.
// A chunk always covers four records: the last chunk of a range reads into the next cell's list (or the padding behind the
// last one) -- triangles nobody asked for, tested all the same; the answer is the minimum over ALL triangles, so a test too
// many cannot change it.  The threshold of the plane / disc bound is reach^2 here (>= TriSearchState::thr: reach >= r (1 + 3e-6),
// thr = r^2 (1 + 3e-6)(1 + 1.1e-6) -- a hair looser, one multiplication instead of a second cross-lane value); an owner's reach
// only changes in a flush, which the whole wave takes together and which republishes it.
template <int BT>
__device__ __forceinline__ void tri_scan_shared(const float *p, const float4 *__restrict__ cell_rec,
                                                const float4 *__restrict__ tri9, TriSearchState &s, float eps_plane,
                                                const int (*seg_j)[BT], const unsigned short (*seg_n)[BT], int &n_seg, unsigned *tab,
                                                TriPool &pool, double delta, float cutf, int *surv = nullptr, int *ev = nullptr,
                                                int *trips = nullptr)
{
    constexpr int Q = TRI_SHARE_Q;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int left = 0;                                                   // chunks this lane still has to hand out
    for (int k = 0; k < n_seg; ++k) left += ((int)seg_n[k][threadIdx.x] + 3) >> 2;
    int k = 0, j = 0, end = 0;
    while (true) {
        const int g = min(Q, left);
        // table order: the lanes' FIRST chunks, then their second ones, ... -- every owner gets its records tested a few at a
        // time, with a flush (and a shorter reach) in between whenever the pool fills, as in the per-lane walk; owner by owner,
        // the first 32 records of an owner would all be tested against the reach it started with (38 survivors per query in
        // the first iteration instead of 23).  One ballot per rank gives the positions.
        int G = 0;
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            const unsigned long long mi = __ballot(i < g);
            if (i < g) {
                if (j >= end) { j = seg_j[k][threadIdx.x]; end = j + (int)seg_n[k][threadIdx.x]; ++k; }
                tab[G + __popcll(mi & lt)] = ((unsigned)j << 6) | (unsigned)lane;
                j += 4;
            }
            G += __popcll(mi);
        }
        if (G == 0) break;
        left -= g;
        const int q = lane & 3, c0 = lane >> 2;
        // 32 chunks = 128 records per step, two records per lane (from two different chunks); the NEXT step's records are
        // fetched before this step's are tested, so that a wave always has loads in flight while it computes (with the
        // four-record trips it had eight loads in flight, then none: the scan is a chain of round trips, and with four waves
        // per SIMD nothing else hides them)
        struct Step { bool act[2]; int own[2]; float4 r0[2], r1[2]; };
        auto fetch = [&](int base, Step &st) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = base + 16 * u + c0;
                st.act[u] = c < G;
                // (no branch around the loads: a lane without a chunk reads record 0 and drops it -- loads behind a branch
                //  make the compiler wait for ALL outstanding loads before the first use, and the overlap is gone)
                const unsigned ent = tab[st.act[u] ? c : 0];
                st.own[u] = st.act[u] ? (int)(ent & 63u) : lane;
                const int rec = st.act[u] ? (int)(ent >> 6) + q : 0;
                st.r0[u] = tri_ld_rec(cell_rec, rec, 0); st.r1[u] = tri_ld_rec(cell_rec, rec, 1);
            }
        };
        auto test = [&](const Step &st) {
            const float4 oa4 = pool.q4[st.own[0]], ob4 = pool.q4[st.own[1]];
            tri_candidate(oa4.x, oa4.y, oa4.z, st.r0[0], st.r1[0], st.act[0], oa4.w, oa4.w * oa4.w, st.own[0], eps_plane, pool, surv);
            tri_candidate(ob4.x, ob4.y, ob4.z, st.r0[1], st.r1[1], st.act[1], ob4.w, ob4.w * ob4.w, st.own[1], eps_plane, pool, surv);
            // (each test adds <= 64 entries; after a flush an owner's reach may be shorter than what later tests read: never longer.
            //  Flushing only when the two tests' survivors would not fit -- test, count, flush, enter -- was measured in round 6:
            //  slower, 0.355 against 0.340 ms per search: the survivors' bounds and flags live across the flush.)
            if (pool.n > TRI_POOL - 128) tri_pool_flush(p, tri9, s, pool, delta, cutf, ev, trips);
        };
        Step A, B;
        fetch(0, A);
        for (int base = 0; base < G; base += 64) {
            fetch(base + 32, B);                                    // (past the end: nothing is loaded)
            test(A);
            fetch(base + 64, A);
            test(B);
        }
    }
    n_seg = 0;
}

// ---- the list of what a front search leaves over (k_tri_settle; the experiment k_tri_accept) ------------------------------------
// Appends are one atomic per wave -- and 15 000 wave atomics on ONE word serialise at ~12 ns each (the whole launch, 190 us,
// when nothing settles).  So the list has ULIST_PARTS parts, each with its own counter (128 bytes apart) and its own region of
// `ulist` (`cap` slots); wave w appends to part w mod ULIST_PARTS.  A consumer adds the counters up and finds item t's part by
// their prefix sums (ulist_item).  counters[0] < 0 says "the list is everybody" (k_tri_settle's gate).
constexpr int ULIST_PARTS = 64, ULIST_STRIDE = 32;                   // (stride in ints)
__host__ __device__ inline int ulist_cap(int n_waves, int per_wave) { return (n_waves + ULIST_PARTS - 1) / ULIST_PARTS * per_wave; }

// called by a whole wave: lanes with `pred` append `item`.  wave_id: the wave's number in the launch
__device__ __forceinline__ void ulist_append(int *__restrict__ ulist, int *__restrict__ counters, int cap, int wave_id, bool pred, int item)
{
    const unsigned long long m = __ballot(pred);
    if (!m) return;
    const int lane = threadIdx.x & 63, part = wave_id % ULIST_PARTS;
    int base = 0;
    if (lane == 0) base = atomicAdd(&counters[part * ULIST_STRIDE], __popcll(m));
    base = __shfl(base, 0, 64);
    if (pred) ulist[(long long)part * cap + base + __popcll(m & ((1ull << lane) - 1ull))] = item;
}

// total length (negative: everybody)
__device__ __forceinline__ int ulist_total(const int *__restrict__ counters)
{
    const int c0 = counters[0];
    if (c0 < 0) return -1;
    int t = c0;
    for (int k = 1; k < ULIST_PARTS; ++k) t += counters[k * ULIST_STRIDE];
    return t;
}

// item t of the list (0 <= t < ulist_total)
__device__ __forceinline__ int ulist_item(const int *__restrict__ ulist, const int *__restrict__ counters, int cap, int t)
{
    int part = 0, before = 0, run = 0;
    for (int k = 0; k < ULIST_PARTS; ++k) {                           // (uniform loads; the compares are the only per-lane work)
        const int c = counters[k * ULIST_STRIDE];
        if (t >= run) { part = k; before = run; }
        run += c;
    }
    return ulist[(long long)part * cap + (t - before)];
}

// ---- seed + neighbours settle a query (round 5; the build and the proof: oa_tri_ring.hpp) -------------------------------------
constexpr int TRI_RING_MAX = 15;          // neighbours listed per triangle: ring[16 t .. 16 t + 14], ring[16 t + 15] = how many
constexpr int TRI_RING_STRIDE = 16;       // ints per triangle (one 64-byte line)

// closest_on_tri (oa_kernels.hpp) for a SEED: the same point, bit for bit, plus `certified` = the barycentric weights the
// point was put together from lie inside the simplex whatever the rounding: a corner; an edge point with its parameter in
// [0, 1] (a quotient of two non-negative floats whose denominator is their sum, or d1 / (d1 - d3) with d1 >= 0 >= d3); or an
// interior point with va, vb, vc all positive.  Then r is within 16 u |coords| of a true point of the triangle.
__device__ __forceinline__ bool tri_seed_certified(const float *p, const float *a, const float *b, const float *c)
{
    const float abx = b[0] - a[0], aby = b[1] - a[1], abz = b[2] - a[2], acx = c[0] - a[0], acy = c[1] - a[1], acz = c[2] - a[2];
    const float apx = p[0] - a[0], apy = p[1] - a[1], apz = p[2] - a[2], bpx = p[0] - b[0], bpy = p[1] - b[1], bpz = p[2] - b[2];
    const float cpx = p[0] - c[0], cpy = p[1] - c[1], cpz = p[2] - c[2];
#define OA_DOT3(ux, uy, uz, vx, vy, vz) (((ux) * (vx) + (uy) * (vy)) + (uz) * (vz))
    const float d1 = OA_DOT3(abx, aby, abz, apx, apy, apz), d2 = OA_DOT3(acx, acy, acz, apx, apy, apz);
    const float d3 = OA_DOT3(abx, aby, abz, bpx, bpy, bpz), d4 = OA_DOT3(acx, acy, acz, bpx, bpy, bpz);
    const float d5 = OA_DOT3(abx, aby, abz, cpx, cpy, cpz), d6 = OA_DOT3(acx, acy, acz, cpx, cpy, cpz);
#undef OA_DOT3
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    const float d43 = d4 - d3, d56 = d5 - d6;
    const bool at_a = d1 <= 0.0f && d2 <= 0.0f;
    const bool at_b = d3 >= 0.0f && d4 <= d3;
    const bool on_ab = vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f;
    const bool at_c = d6 >= 0.0f && d5 <= d6;
    const bool on_ac = vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f;
    const bool on_bc = va <= 0.0f && d43 >= 0.0f && d56 >= 0.0f;
    // (the edge parameters: numerator >= 0 and denominator = numerator + something >= 0, so the quotient is in [0, 1] or NaN --
    //  and a NaN point never becomes anybody's best)
    if (at_a || at_b || on_ab || at_c || on_ac || on_bc) return true;
    return va > 0.0f && vb > 0.0f && vc > 0.0f;
}

// the test of the header.  deltaf >= the search's delta (64 u (scale + |p|_1) + slack)
__device__ __forceinline__ bool tri_ring_accepts(float d2, float accept, float deltaf)
{
    const float tt = accept - 2.5f * deltaf;
    return tt > 0.f && d2 < tt * tt * 0.99998f;                     // (sqrt(d2) (1 + 1e-5) < tt, squared and rounded down)
}

// The seed + its neighbours for EVERY query of the shard, one thread each, and nothing else: what they settle gets its answer
// here (keys[]: (bits(d2) << 32) | triangle, the searches' own format), what they do not goes on a list that k_tri_search_grid
// works through (qlist).  Two launches instead of one because the grid search is built around its wave: listing cells, the
// shared scan, the pool and its flushes cost a wave nearly the same whether 3 or 64 of its lanes still search (measured: with
// 95 % of the queries settled in the search's own prologue a launch took what it took before) -- and its 40 KB of LDS per
// workgroup cap the chip at four such waves per SIMD.  Here: no LDS, half the registers; the list packs the rest densely.
// The list: ulist_* above (this launch zeroes the NEXT launch's counters: the host alternates between two sets).
#if !defined(OA_FAMILY_TU) && defined(OA_EXPERIMENTS)      // experiment: only in liboa_icp_exp.so's host translation unit
__global__ __launch_bounds__(256) void k_tri_accept(const DevState *__restrict__ st, const float4 *__restrict__ src4, int ns, float scalef,
                                                    const float4 *__restrict__ tri9, const int *__restrict__ prev,
                                                    const int *__restrict__ ring, unsigned long long *__restrict__ keys,
                                                    int *__restrict__ ulist, int ulist_cap_, int *__restrict__ counters, int *__restrict__ counters_next)
{
    if (st->halt) return;
    if (blockIdx.x == 0 && threadIdx.x < ULIST_PARTS) counters_next[threadIdx.x * ULIST_STRIDE] = 0;
    const int vb = xcd_block_index();
    const int i = vb * (int)blockDim.x + threadIdx.x;
    const bool alive = i < ns;
    bool accepted = false;
    float best = INFINITY;
    uint32_t bidx = IDX_NONE;
    float pf[3] = { 0.f, 0.f, 0.f };
    int s = -1;
    if (alive) {
        const float4 p4 = src4[i];
        co_find(st, p4.x, p4.y, p4.z, pf[0], pf[1], pf[2]);
        s = prev[i];
    }
    if (s >= 0) {
        float a[3], b[3], c[3], r[3];
        const float4 u = tri9[3ll * s], v = tri9[3ll * s + 1], w = tri9[3ll * s + 2];
        a[0] = u.x; a[1] = u.y; a[2] = u.z; b[0] = u.w; b[1] = v.x; b[2] = v.y; c[0] = v.z; c[1] = v.w; c[2] = w.x;
        closest_on_tri(pf, a, b, c, r);
        const float d = tri_dist2(pf, r);
        if (d < INFINITY) { best = d; bidx = (uint32_t)s; }
        const float deltaf = (fabsf(pf[0]) + fabsf(pf[1]) + fabsf(pf[2]) + scalef) * 3.8148e-6f + scalef * 1.1e-10f;
        accepted = d < INFINITY && tri_ring_accepts(d, w.y, deltaf);
        if (accepted) accepted = tri_seed_certified(pf, a, b, c);
    }
    if (__any(accepted)) {
        const int4 *rp = (const int4 *)(ring + (size_t)TRI_RING_STRIDE * (size_t)(accepted ? s : 0));
        const int cnt = accepted ? rp[3].w : 0;
        int4 m = rp[0];
        for (int k0 = 0; __any(k0 < cnt); k0 += 4) {
            const int4 mn = rp[min((k0 >> 2) + 1, 3)];               // the next four, in flight during these
            const int id[4] = { m.x, m.y, m.z, m.w };
            float4 tu[4], tv[4], tw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // (a lane that is through reads triangle 0 and drops it: no branch around the loads)
                const long long tt = (k0 + j < cnt) ? (long long)id[j] : 0ll;
                tu[j] = tri9[3 * tt]; tv[j] = tri9[3 * tt + 1]; tw[j] = tri9[3 * tt + 2];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j < cnt) {
                    const float a[3] = { tu[j].x, tu[j].y, tu[j].z }, b[3] = { tu[j].w, tv[j].x, tv[j].y }, c[3] = { tv[j].z, tv[j].w, tw[j].x };
                    float r[3];
                    closest_on_tri(pf, a, b, c, r);
                    const float d = tri_dist2(pf, r);
                    const uint32_t t = (uint32_t)id[j];
                    if (d < best || (d == best && t < bidx && d < INFINITY)) { best = d; bidx = t; }
                }
            }
            m = mn;
        }
    }
    if (accepted) keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    // the others, packed: one atomic per wave
    ulist_append(ulist, counters, ulist_cap_, vb * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6), alive && !accepted, i);
}
#endif  // !OA_FAMILY_TU

// L = 1, 2 or 4 lanes per query, as in k_nn_search_grid: the rows of a ring are dealt out to the lanes, every lane runs
// both phases on its rows with its own state, and the lanes merge (d2, index) after every batch of rows.
// STATS (debug builds of the launch, OA_GRID_STATS=1): per-launch totals of what the queries did, see TRI_STAT_*
enum { TRI_STAT_QUERIES, TRI_STAT_ROWS, TRI_STAT_ENTRIES, TRI_STAT_SURVIVORS, TRI_STAT_EVALS, TRI_STAT_WAVE_TRIPS,
       TRI_STAT_WAVE_MAX_ENTRIES, TRI_STAT_WAVE_MAX_ROWS, TRI_STAT_RING2, TRI_STAT_RING3, TRI_STAT_UNSETTLED, TRI_STAT_OVER, TRI_STAT_SPHERE,
       TRI_STAT_WAVES, TRI_STAT_CYC_TOTAL, TRI_STAT_CYC_PROLOGUE, TRI_STAT_CYC_LIST, TRI_STAT_CYC_SCAN, TRI_STAT_CYC_FLUSH, TRI_STAT_CYC_BOOK,
       TRI_STAT_LOOP_TRIPS, TRI_STAT_N };

// ACC (the loop's iterations, round 4 -- as k_nn_search_grid<L, true>): ONE launch does what took three.  (i) A query the
// rings did not settle is finished through the triangle tree right here, by the wave that owns it (bvh_wave_query<true>,
// seeded with the grid's partial answer exactly as k_bvh_search seeds it from keys[]; its scratch lies over the wave's
// range lists, which are dead by then).  (ii) The closest point on the winning triangle, the pair test and the iteration's
// fp64 sums are taken in the epilogue: one row of `partials` per workgroup, the rows k_pair_accumulate_canon writes for the
// same shard, bit for bit (the host switches between the two from one iteration to the next, oa_icp.hip: grid_fast_now).
// prev[] gets the winner (the next search's seed); keys[] is not written: nothing reads it inside the loop.
#ifndef OA_TRI_LIST_REPS
#define OA_TRI_LIST_REPS 3
#endif
template <int L, bool STATS = false, bool SHARE = true, bool ACC = false, int BT = 256>
#ifndef OA_TRI_MIN_WAVES
#define OA_TRI_MIN_WAVES 4
#endif
__global__ __launch_bounds__(BT, OA_TRI_MIN_WAVES) void k_tri_search_grid(const DevState *__restrict__ st,
                                                         const float4 *__restrict__ src4, int ns, GridParams gp,
                                                         const int *__restrict__ cell_start,
                                                         const float4 *__restrict__ cell_rec,
                                                         const float4 *__restrict__ tri9,
                                                         int *__restrict__ prev,
                                                         unsigned long long *__restrict__ keys,
                                                         int *__restrict__ todo_list, int *__restrict__ todo_count, int turn,
                                                         unsigned long long *__restrict__ stats = nullptr,
                                                         BvhParams bp = BvhParams{}, const float4 *__restrict__ boxes = nullptr,
                                                         const float4 *__restrict__ prims = nullptr, NormalTest nrm = NormalTest{},
                                                         double *__restrict__ partials = nullptr,
                                                         const int *__restrict__ ring = nullptr,
                                                         const int *__restrict__ qlist = nullptr, const int *__restrict__ qcount = nullptr,
                                                         int qmin = 0, int qmax = 0x7FFFFFFF, int qcap = 0)
{
    // rows per lane and batch: nine rows of a ring at a time (dealt out to the L lanes of the query).  The rows of the
    // first block of a search are whole ranges (one per row); interior rows of later rings contribute their two end
    // cells (two ranges per row): with one lane per query a batch then takes as many rows as fit the range list.
    constexpr int RPL = (9 + L - 1) / L;
    static_assert(RPL <= TRI_SEGS && (L == 1 || 2 * RPL <= TRI_SEGS), "a batch of rows must fit the per-thread range list");
    if (st->halt) return;
    if (turn >= 0 && (st->tree_turn != 0) != (turn != 0)) return;  // not this kernel's turn (DevState::tree_turn)
    // BT = 64 (round 6, the plain searches of long loops): ONE wave per workgroup.  The waves of this search share nothing, but a
    // workgroup's slot is only free again when its slowest wave is through.
    static_assert(BT == 256 || (BT == 64 && !ACC), "the accumulating search's rows of partials are workgroups of 256 queries");
    __shared__ unsigned pool_to[BT / 64][TRI_POOL];
    __shared__ unsigned short pool_key[BT / 64][TRI_POOL];
    __shared__ unsigned long long pool_slot[BT], pool_first[BT];
    __shared__ __attribute__((aligned(16))) int seg_j[TRI_SEGS][BT];
    __shared__ __attribute__((aligned(16))) unsigned short seg_n[TRI_SEGS][BT];                 // (a range is never longer than the budget it was charged to: < 65536, build_tri_grid)
    __shared__ unsigned chunk_tab[SHARE ? BT / 64 : 1][SHARE ? 64 * TRI_SHARE_Q : 1];
    __shared__ float4 owner_q4[SHARE ? BT : 1];
    int n_seg = 0;
    TriPool pool;
    {
        const int wv = threadIdx.x >> 6;
        pool.to = pool_to[wv]; pool.key = pool_key[wv]; pool.slot = pool_slot + 64 * wv;
        pool.first = pool_first + 64 * wv;
        pool.q4 = SHARE ? owner_q4 + 64 * wv : nullptr;
        pool.n = 0;
        pool_first[threadIdx.x] = ~0ull;                            // (this wave's own words: no barrier needed)
    }
    int n_rows_loaded = 0, n_entries = 0, n_surv = 0, n_evals = 0, n_trips = 0, max_ring = 0;
    // (instrumented build) shader-clock stamps at the phase boundaries, summed per wave
    long long cyc_t0 = 0, cyc_mark = 0, cyc_list = 0, cyc_scan = 0, cyc_flush = 0, cyc_book = 0, cyc_prologue = 0;
    int n_loop_trips = 0;
    if (STATS) cyc_t0 = cyc_mark = (long long)__builtin_readcyclecounter();
#define OA_TRI_STAMP(acc) do { if (STATS) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - cyc_mark; cyc_mark = now_; } } while (0)
    // qlist (not with ACC): the queries are the list (ulist_*) a front search -- k_tri_settle -- left over, in the dispatcher's own
    // workgroup order (the list is short: an XCD's contiguous share of the launch would leave seven XCDs idle)
    bool listed = !ACC && qlist != nullptr;
    if (listed) {
        // The list's length is only known on the device, and it decides how many lanes a query should get (a short list leaves
        // the chip's wave slots empty: then the rows of a ring are dealt out to 4 lanes): the host enqueues one launch per
        // regime and each runs only when the length is in ITS range (qmin, qmax] -- an empty launch costs a few microseconds.
        // A NEGATIVE length says "everybody" (k_tri_settle's gate): the launch sized for the shard runs as if it had no list.
        const int qn = ulist_total(qcount);
        if (qn < 0) { if (qmax != 0x7FFFFFFF) return; listed = false; }
        else { ns = qn; if (ns <= qmin || ns > qmax) return; }
    }
    // one contiguous part of the queries per XCD -- in chunks while the pose still moves (xcd_block_index_chunked)
    const bool moving = pose_moving(st, gp.scale, gp.moving_h);
    const int vb = listed ? (int)blockIdx.x : ((gp.xcd_chunk > 0 && pose_moving(st, gp.scale, gp.xcd_moving_h)) ? xcd_block_index_chunked(gp.xcd_chunk * (256 / BT)) : xcd_block_index());
    const int gt = vb * (int)blockDim.x + threadIdx.x;
    int i = gt / L;
    const int sub = gt % L;                                         // the L lanes of a query are neighbours in a wave
    if (listed && (vb * (int)blockDim.x) / L >= ns) return;         // (the whole workgroup: before any barrier)
    // Lanes past the last query stay: phase 2 deals pool entries to ALL 64 lanes of the wave (entry e to lane e mod 64),
    // so a lane that left would take its share of the entries with it.  They repeat the last query without any say.
    const bool alive = i < ns;
    if (!alive) i = ns - 1;
    if (listed) i = ulist_item(qlist, qcount, qcap, i);             // from here on: the query's slot
    const float4 p4 = src4[i];
    float pf[3];
    co_find(st, p4.x, p4.y, p4.z, pf[0], pf[1], pf[2]);          // co_find (general.py:287)

    float best = INFINITY;
    uint32_t bidx = IDX_NONE;
    const int s = prev ? prev[i] : -1;
    // Seed + neighbours (oa_tri_ring.hpp): a query within A(seed) of its seed is settled by the seed and the <= TRI_RING_MAX
    // triangles listed beside it -- the minimum over ALL triangles is among them; no cell is listed, no record scanned.
    bool accepted = false;
    const float ring_scalef = (float)gp.scale * 1.000001f;
    if (s >= 0) {
        float a[3], b[3], c[3], r[3];
        const float4 u = tri9[3ll * s], v = tri9[3ll * s + 1], w = tri9[3ll * s + 2];
        a[0] = u.x; a[1] = u.y; a[2] = u.z; b[0] = u.w; b[1] = v.x; b[2] = v.y; c[0] = v.z; c[1] = v.w; c[2] = w.x;
        closest_on_tri(pf, a, b, c, r);
        const float d = tri_dist2(pf, r);
        if (d < INFINITY) { best = d; bidx = (uint32_t)s; }
        if (ring) {
            // (delta of the search below, rounded up: 64 u (scale + |p|_1) + slack)
            const float deltaf = (fabsf(pf[0]) + fabsf(pf[1]) + fabsf(pf[2]) + ring_scalef) * 3.8148e-6f + ring_scalef * 1.1e-10f;
            accepted = d < INFINITY && tri_ring_accepts(d, w.y, deltaf);
            if (accepted) accepted = tri_seed_certified(pf, a, b, c);
        }
    }
    if (ring && __any(accepted)) {
        const int4 *rp = (const int4 *)(ring + (size_t)TRI_RING_STRIDE * (size_t)(accepted ? s : 0));
        const int cnt = accepted ? rp[3].w : 0;
        int4 m = rp[0];
        for (int k0 = 0; __any(k0 < cnt); k0 += 4) {
            const int4 mn = rp[min((k0 >> 2) + 1, 3)];               // the next four, in flight during these
            const int id[4] = { m.x, m.y, m.z, m.w };
            float4 tu[4], tv[4], tw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // (a lane that is through reads triangle 0 and drops it: no branch around the loads)
                const long long tt = (k0 + j < cnt) ? (long long)id[j] : 0ll;
                tu[j] = tri9[3 * tt]; tv[j] = tri9[3 * tt + 1]; tw[j] = tri9[3 * tt + 2];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (k0 + j < cnt) {
                    const float a[3] = { tu[j].x, tu[j].y, tu[j].z }, b[3] = { tu[j].w, tv[j].x, tv[j].y }, c[3] = { tv[j].z, tv[j].w, tw[j].x };
                    float r[3];
                    closest_on_tri(pf, a, b, c, r);
                    const float d = tri_dist2(pf, r);
                    const uint32_t t = (uint32_t)id[j];
                    if (d < best || (d == best && t < bidx && d < INFINITY)) { best = d; bidx = t; }
                }
            }
            m = mn;
        }
    }

    double pabs = 0.0;
    const GridQuery q = grid_locate(gp, pf[0], pf[1], pf[2], &pabs);
    const float h = gp.hf, inv_h = gp.inv_hf, slack = gp.slackf;
    // float32 closest-point evaluation can undershoot the real distance by at most delta
    const double delta = 64.0 * 5.9604644775390625e-08 * (gp.scale + pabs) + gp.slack;
    // `lim`: the best so far or the search radius (search_cutoff2), whichever is smaller -- see k_nn_search_grid
    const float cutf = search_cutoff2(st, pf[0], pf[1], pf[2]);
    TriSearchState S;
    S.best = best; S.bidx = bidx;
    S.lim = fminf(best, cutf);
    tri_state_refresh(S, delta);
    if (SHARE) owner_q4[threadIdx.x] = make_float4(pf[0], pf[1], pf[2], S.reach);
    bool settled = false, over = false;
    // candidates this query may look at (see GridParams; split between its lanes).  While the pose still moves by a good part of a cell per
    // iteration (first iteration, or last iteration's translation + rotation x object size above h / 4) the seeds are
    // stale and most queries need the second ring: handing them all to the tree costs more than letting the grid
    // look at twice as many candidates.
    int budget = gp.budget;
    int budget_extra = 0;                                            // what the larger budget added (wave-uniform)
    {
        if (moving) budget = gp.budget_moving;
        if (L > 1) budget = budget / L + 8;
        budget_extra = budget - (L > 1 ? gp.budget / L + 8 : gp.budget);
    }
    const int r_start = (S.bidx != IDX_NONE && gp.seeded_start) ? 1 : 0;   // as in k_nn_search_grid
    // ONE wave-uniform loop: every trip, every lane that still has work lists the cell ranges of its next batch of rows
    // (its own ring, its own batch), then the whole wave scans and flushes together.  The pool is shared by the wave,
    // so phase 1 and phase 2 must be reached by all of its lanes at the same time -- hence no per-lane loops around
    // them: a lane's ring / batch counters are plain state, and the only loop condition is __any(busy).
    int r = r_start, b0 = 0;
    bool busy = q.finite && alive && !accepted;
    settled = accepted;
    OA_TRI_STAMP(cyc_prologue);
    while (__any(busy)) {
        bool ring_done = false;
        if (STATS) ++n_loop_trips;
        if (busy && r == 1) {
            // the 3 x 3 block, as in k_nn_search_grid (GridBlock3): which rows can matter -- the row test of the general code
            // below -- then their ranges, UR rows in flight at a time
            GridBlock3 blk = grid_block3_select<L>(gp, q, sub, r_start == 1, S.reach2f, 0.f, S.reach2f, 0.999998f);
            if (STATS) n_rows_loaded += __popc(blk.rows);
#ifndef OA_TRI_BLOCK3_UR
#define OA_TRI_BLOCK3_UR grid_block3_unroll(L)
#endif
            constexpr int UR = OA_TRI_BLOCK3_UR;
            while (blk.rows) {
                int ja[UR], jb[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) grid_block3_next(gp, q, cell_start, blk, ja[u], jb[u]);
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    if (jb[u] > ja[u] && budget >= 0) {
                        budget -= jb[u] - ja[u];                         // crowded cells: one wave of the tree search is faster
                        if (budget >= 0) { seg_j[n_seg][threadIdx.x] = ja[u]; seg_n[n_seg][threadIdx.x] = (unsigned short)(jb[u] - ja[u]); ++n_seg; if (STATS) n_entries += jb[u] - ja[u]; }
                    }
                }
            }
            ring_done = true;
        } else if (busy) {
            constexpr int GR = RPL;                                 // rows per lane and batch here (r = 0, r >= 2)
            const bool first = (r == r_start);
            const int side = 2 * r + 1, n_rows = side * side;
            const unsigned div_mul = 65536u / (unsigned)side + 1u;  // k / side == (k * div_mul) >> 16 for k < 256, side <= 15
            const int rpl = GR;
            // Several batches of rows per trip while the range list has room (round 6): far from the surface most rows of a ring are
            // empty, and a trip -- the wave's scan and flush -- per nine rows made a ring-3 query take six trips for a handful of ranges
            // (L == 1 takes as many rows of a batch as fit the list; with several lanes per query a batch goes in whole or not at all)
            for (int rep = 0; rep < OA_TRI_LIST_REPS && !ring_done && n_seg + (L == 1 ? 2 : 2 * RPL) <= TRI_SEGS && budget >= 0; ++rep) {
            // rows of the ring: cell ranges first (independent loads), then the candidates -- as in k_nn_search_grid;
            // per-row arithmetic in float on the query's frame (GridQuery)
            int ja[GR], jb[GR], jc[GR], jd[GR];
#pragma unroll
            for (int k = 0; k < GR; ++k) {
                ja[k] = jb[k] = 0;
                jc[k] = jd[k] = 0;
                const int kk = b0 + sub + L * k;
                if (k >= rpl || kk >= n_rows) continue;
                const int qz = (int)(((unsigned)kk * div_mul) >> 16);
                const int dzi = qz - r, dyi = kk - qz * side - r;
                const int z = q.c[2] + dzi, y = q.c[1] + dyi;
                if (z < 0 || z >= gp.n[2] || y < 0 || y >= gp.n[1]) continue;
                const float gz = grid_gap(q.f[2], h, slack, dzi), gy = grid_gap(q.f[1], h, slack, dyi);
                const float row2 = __builtin_fmaf(gy, gy, __builtin_fmaf(gz, gz, q.off2));
                // (sqrt(row2) - delta)^2 (1 - 1e-5) - 1e-30 > lim  <=>  row2 > reach2: nothing in this row can matter
                if (row2 * 0.999998f > S.reach2f) continue;
                // cells of the row whose slab along x can still hold a triangle within reach
                int dl, dr;
                grid_row_span(q.f[0], h, inv_h, slack, S.reach2f - row2 * 0.999998f, r, dl, dr);
                const int xa = max(q.c[0] - dl, 0), xb = min(q.c[0] + dr, gp.n[0] - 1);
                const int row = (z * gp.n[1] + y) * gp.n[0];
                // interior rows were fully covered by ring r-1: only their two end cells are new
                const bool shell_row = first || dzi == -r || dzi == r || dyi == -r || dyi == r;
                if (STATS) ++n_rows_loaded;
                if (shell_row) {
                    ja[k] = cell_start[row + xa]; jb[k] = cell_start[row + xb + 1];
                } else {
                    const int xl = q.c[0] - r, xr = q.c[0] + r;
                    if (dl == r && xl >= 0) { ja[k] = cell_start[row + xl]; jb[k] = cell_start[row + xl + 1]; }
                    if (dr == r && xr < gp.n[0]) { jc[k] = cell_start[row + xr]; jd[k] = cell_start[row + xr + 1]; }
                }
            }
            // The non-empty cell-list ranges of this lane's rows go on a per-thread list in LDS and are scanned in ONE
            // flattened loop (tri_scan_segments): every lane walks its own ranges four records per trip, so a wave
            // pays max-over-lanes(records) trips.  Walking the rows in lockstep instead -- range m of every lane
            // together -- pays the sum over m of the longest range m, several times more when cells hold ~20 records
            // (PMC before: 41k VALU instructions per wave in the first iterations of a run).
            int consumed = rpl;                                   // rows of this batch that went on the list (L == 1: as many as fit)
            bool full = false;
#pragma unroll
            for (int k = 0; k < GR; ++k) {
                if (L == 1 && !full && n_seg + 2 > TRI_SEGS) { full = true; consumed = k; }
                if (full) continue;
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    const int j0 = sg ? jc[k] : ja[k], j1 = sg ? jd[k] : jb[k];
                    if (j1 > j0 && budget >= 0) {
                        budget -= j1 - j0;                           // crowded cells: one wave of the tree search is faster
                        if (budget >= 0) { seg_j[n_seg][threadIdx.x] = j0; seg_n[n_seg][threadIdx.x] = (unsigned short)(j1 - j0); ++n_seg; if (STATS) n_entries += j1 - j0; }
                    }
                }
            }
            b0 += consumed * L;
            ring_done = b0 >= n_rows;
            }
        }
        // over its budget: the tree takes the query anyway -- with the seed's bound, which the records scanned so far rarely
        // improve on -- so what this batch listed is not scanned (crowded cells: up to `budget` records per lane for nothing)
        if (gp.drop_over && busy && budget < 0) n_seg = 0;
        OA_TRI_STAMP(cyc_list);
        // phase 1 and phase 2: the whole wave, every trip
        if (SHARE) tri_scan_shared(pf, cell_rec, tri9, S, gp.eps_plane, seg_j, seg_n, n_seg, chunk_tab[threadIdx.x >> 6], pool, delta, cutf,
                                   STATS ? &n_surv : nullptr, STATS ? &n_evals : nullptr, STATS ? &n_trips : nullptr);
        else tri_scan_segments(pf, cell_rec, tri9, S, gp.eps_plane, seg_j, seg_n, n_seg, pool, delta, cutf,
                               STATS ? &n_surv : nullptr, STATS ? &n_evals : nullptr, STATS ? &n_trips : nullptr);
        // phase 2 now if somebody needs its final word on this ring (or gives up), or the pool is filling up; otherwise the
        // survivors wait for the next batch's (every flush costs the wave at least one evaluation trip)
        OA_TRI_STAMP(cyc_scan);
        if (__any(busy && (ring_done || budget < 0)) || pool.n > TRI_POOL / 4)          // (64 ... 304 measured in round 6: no difference)
            tri_pool_flush(pf, tri9, S, pool, delta, cutf, STATS ? &n_evals : nullptr, STATS ? &n_trips : nullptr);
        OA_TRI_STAMP(cyc_flush);
        over = busy && budget < 0;
        if (L > 1) {                                             // the lanes of the query agree on the best so far
            bool changed = false;
#pragma unroll
            for (int o = 1; o < L; o <<= 1) {
                const float ob = __shfl_xor(S.best, o, 64);
                const uint32_t oi = (uint32_t)__shfl_xor((int)S.bidx, o, 64);
                if (ob < S.best) { S.best = ob; S.bidx = oi; changed = true; }
                else if (ob == S.best && oi < S.bidx) S.bidx = oi;
                over = (__shfl_xor((int)over, o, 64) != 0) || over;
            }
            if (changed) { S.lim = fminf(S.best, cutf); tri_state_refresh(S, delta); if (SHARE) owner_q4[threadIdx.x].w = S.reach; }
        }
        if (busy) {
            if (over) busy = false;
            else if (ring_done) {
                if (STATS) max_ring = r;
                const float bound = grid_cube_bound2(gp, q, r);        // everything outside the cube of radius r
                if (!(bound < INFINITY) || bound * 0.999998f > S.reach2f) { settled = true; busy = false; }   // same test as for a row
                else { ++r; b0 = 0; if (r > gp.r_max) busy = false; }
            }
        }
        OA_TRI_STAMP(cyc_book);
    }
#undef OA_TRI_STAMP
    if (STATS && stats) {
        // per-wave totals first (shuffles), ONE atomic per counter and wave: a million threads adding to the same dozen words
        // at the end of the launch slow the waves that are still searching
        const long long cyc_total = (long long)__builtin_readcyclecounter() - cyc_t0;
        int sv[10] = { alive ? 1 : 0, alive ? n_rows_loaded : 0, alive ? n_entries : 0, n_surv & 0xFFFF, (int)((unsigned)n_surv >> 16), n_evals,
                       alive && max_ring >= 2, alive && max_ring >= 3, alive && !settled, alive && over };
#pragma unroll
        for (int k = 0; k < 10; ++k)
            for (int o = 32; o > 0; o >>= 1) sv[k] += __shfl_xor(sv[k], o, 64);
        int me = n_entries, mr = n_rows_loaded;
        for (int o = 32; o > 0; o >>= 1) { me = max(me, __shfl_xor(me, o, 64)); mr = max(mr, __shfl_xor(mr, o, 64)); }
        if ((threadIdx.x & 63) == 0) {
            // one row of counters per wave, plain stores (the host adds the rows up)
            unsigned long long *row = stats + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TRI_STAT_N;
            const int keys[10] = { TRI_STAT_QUERIES, TRI_STAT_ROWS, TRI_STAT_ENTRIES, TRI_STAT_SURVIVORS, TRI_STAT_SPHERE, TRI_STAT_EVALS,
                                   TRI_STAT_RING2, TRI_STAT_RING3, TRI_STAT_UNSETTLED, TRI_STAT_OVER };
#pragma unroll
            for (int k = 0; k < 10; ++k) row[keys[k]] = (unsigned long long)sv[k];
            row[TRI_STAT_WAVE_TRIPS] = (unsigned long long)n_trips;
            row[TRI_STAT_WAVE_MAX_ENTRIES] = (unsigned long long)me;
            row[TRI_STAT_WAVE_MAX_ROWS] = (unsigned long long)mr;
            row[TRI_STAT_WAVES] = sv[0] ? 1ull : 0ull;
            row[TRI_STAT_CYC_TOTAL] = (unsigned long long)cyc_total;
            row[TRI_STAT_CYC_PROLOGUE] = (unsigned long long)cyc_prologue;
            row[TRI_STAT_CYC_LIST] = (unsigned long long)cyc_list;
            row[TRI_STAT_CYC_SCAN] = (unsigned long long)cyc_scan;
            row[TRI_STAT_CYC_FLUSH] = (unsigned long long)cyc_flush;
            row[TRI_STAT_CYC_BOOK] = (unsigned long long)cyc_book;
            row[TRI_STAT_LOOP_TRIPS] = (unsigned long long)n_loop_trips;
        }
    }
    // how crowded the hand-over is per wave, against the BASE budget (see k_nn_search_grid): what the host looks at before
    // it lets a later search finish its own leftovers
    bool tight = budget < budget_extra;
    if (L > 1) {
#pragma unroll
        for (int o = 1; o < L; o <<= 1) tight = (__shfl_xor((int)tight, o, 64) != 0) || tight;
    }
    const bool mine = sub == 0 && alive;
    if (!ACC) {
        const unsigned long long crowd = __ballot(mine && q.finite && (!settled || tight));
        if (crowd && (threadIdx.x & 63) == 0) atomicMax(todo_count + 1, __popcll(crowd));
        if (!mine) return;
        keys[i] = ((unsigned long long)__float_as_uint(S.best) << 32) | S.bidx;
        if (!settled) todo_list[atomicAdd(todo_count, 1)] = i;
        return;
    }
    // ---- ACC: finish, record, accumulate -- all threads stay to the end (wave-wide descents, workgroup-wide reduction)
    {
        unsigned long long todo = __ballot(mine && !settled && q.finite);
        const unsigned long long crowd = __ballot(mine && q.finite && (!settled || tight));
        if (crowd && (threadIdx.x & 63) == 0) atomicMax(todo_count + 1, __popcll(crowd));
        if (todo) {
            if ((threadIdx.x & 63) == 0) atomicAdd(todo_count, __popcll(todo));
            // this wave's columns of the range lists are free now: per level 256 B of bounds (seg_j), the mask and the node (seg_n)
            const int lane = threadIdx.x & 63, col0 = threadIdx.x & ~63;
            const BvhLds lds{ (float *)&seg_j[0][col0], BT, (unsigned long long *)&seg_n[0][col0], BT / 4, (int *)((char *)&seg_n[0][col0] + 8), BT / 2 };
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const float qp[3] = { __shfl(pf[0], l, 64), __shfl(pf[1], l, 64), __shfl(pf[2], l, 64) };
                float b = __shfl(S.best, l, 64);
                uint32_t bi = (uint32_t)__shfl((int)S.bidx, l, 64);
                float tx = 0.f, ty = 0.f, tz = 0.f;
                const uint32_t bi0 = bi;
                bvh_wave_query<true>(bp, boxes, prims, qp, __shfl(cutf, l, 64), b, bi, tx, ty, tz, lds, lane);
                if (lane == l && bi != bi0) { S.best = b; S.bidx = bi; }
            }
        }
    }
    bool valid = false;
    float vbx = 0.f, vby = 0.f, vbz = 0.f;
    double dist = 0.0;
    if (mine) {
        prev[i] = (S.bidx == IDX_NONE) ? -1 : (int)S.bidx;           // the next search's seed
        if (S.bidx != IDX_NONE) {
            float ta[3], tb[3], tc[3], rr[3], tn[3] = { 0.f, 0.f, 0.f };
            load_tri(tri9, S.bidx, ta, tb, tc);
            closest_on_tri(pf, ta, tb, tc, rr);
            if (nrm.src_n) {                                         // geometric face normal (Blender normal_tri_v3 order)
                const float e1[3] = { ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2] };
                const float e2[3] = { tb[0] - tc[0], tb[1] - tc[1], tb[2] - tc[2] };
                tn[0] = e1[1] * e2[2] - e1[2] * e2[1];
                tn[1] = e1[2] * e2[0] - e1[0] * e2[2];
                tn[2] = e1[0] * e2[1] - e1[1] * e2[0];
            }
            valid = pair_eval(st, pf[0], pf[1], pf[2], rr[0], rr[1], rr[2], nrm, i, tn, st->thresh, vbx, vby, vbz, dist);
        }
    }
    const double pvx = st->pivot[0], pvy = st->pivot[1], pvz = st->pivot[2];
    const float4 a4 = mine ? src4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();                                                // every wave is through with its range lists: the reduction's scratch lies over them
    block_store_pair(valid, (double)a4.x - pvx, (double)a4.y - pvy, (double)a4.z - pvz, (double)vbx - pvx, (double)vby - pvy,
                     (double)vbz - pvz, dist - st->d_pivot, (double (*)[NSUMS])&seg_j[0][0], partials + (long long)vb * NSUMS);
}

// brute force over all triangles for every source point (OA_SEARCH_BRUTE; the oracle's oo_nn_tri_brute on the device)
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(256) void k_tri_search_all(const DevState *__restrict__ st,
                                                        const float4 *__restrict__ src4, int ns,
                                                        const float4 *__restrict__ tri9, int n_tris,
                                                        const int *__restrict__ prev,
                                                        unsigned long long *__restrict__ keys)
{
    if (st->halt) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const float4 p4 = src4[i];
        float pf[3];
        co_find(st, p4.x, p4.y, p4.z, pf[0], pf[1], pf[2]);
        float best = INFINITY;
        uint32_t bidx = IDX_NONE;
        if (prev && prev[i] >= 0) tri_eval(pf, tri9, (uint32_t)prev[i], best, bidx);
        for (int t = 0; t < n_tris; ++t) tri_eval(pf, tri9, (uint32_t)t, best, bidx);
        keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    }
}
#endif  // !OA_FAMILY_TU

#endif  // __HIPCC__
}  // namespace oa
