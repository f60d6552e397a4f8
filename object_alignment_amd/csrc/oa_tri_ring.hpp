// oa_tri_ring.hpp -- surface mode: when the SEED TRIANGLE AND ITS NEIGHBOURS settle a query (round 5; the surface analogue of
// the vertex grid's safe radii, oa_grid.hpp).
//
// Once the pose has settled, a query's nearest triangle is last iteration's (its seed) or one that touches it, and the query
// lies a small fraction of a triangle away from the surface.  The grid search still walks the cell lists around the query
// (~28 records, pool, flushes) to PROVE that nothing else is nearer.  That proof does not depend on the query: per triangle T
// this build lists
//     ring(T)   the triangles within `tau` of T (all that touch it: lower bound 0), at most TRI_RING_MAX of them, and
//     C(T)      a LOWER bound of the distance from T to every triangle outside ring(T)            (0 = "never", see below).
// A query p whose seed T evaluates (float32, the search's own arithmetic) to the point r at squared distance d2 knows, for every
// T' outside the ring and x in T within eps_r of r:  dist(p, T') >= dist(x, T') - |p - x| >= C - eps_r - |p - r|.  The float32
// evaluation of T' is >= (dist(p, T') - delta)^2 (1 - 1e-5) (oa_tri.hpp: tri_reach_bound), so with
//     A(T) = C(T) / 2 (1 - 1e-4)          stored beside the triangle, tri9[3 t + 2].y
//     sqrt(d2) (1 + 1e-5) + 2.5 delta < A(T)                     (tri_ring_accepts)
// every T' outside the ring evaluates STRICTLY above d2 >= the best of {T} + ring(T): brute force over all triangles would
// report exactly the lexicographic minimum (d2, index) over {T} + ring(T) -- <= 1 + TRI_RING_MAX closest-point evaluations and
// no cell walk.  eps_r: tri_seed_certified() accepts only evaluations whose barycentric weights are provably inside the
// simplex (then r is within 16 u |coords| < delta / 4 of a true point of T, whatever the conditioning of T).
// Everything else -- no seed, a seed whose C is 0, a query farther than A(T) -- takes the rings of cells as before, so bits
// cannot change (tests/test_gpu_tri_ring.py; the fuzzers run with the rings on).
//
// The lower bound of dist(T, T'): for a unit vector u, |x - y| >= u . (y - x) >= min_{T'} u . y - max_T u . x.  Seven axes of T:
// its normal (both signs), the three in-plane outward edge normals and the three directions centroid -> corner.  In a plane the
// six in-plane directions are 60 degrees apart for an equilateral T: the bound is >= cos 30 = 0.87 of the true distance, and exact
// for a regular lattice's second ring (closest features are corner-edge pairs along exactly these directions).  Evaluated in
// double on the float corners (rounding 1e-15 |coords|, covered by the 1e-4 in A); T' outside the cells the build scans has a
// bounding box disjoint from bbox(T) grown by `cap`, hence C = min(cap, the smallest bound seen).
// C(T) = 0 ("never"): degenerate or NaN triangles, more than TRI_RING_MAX neighbours (fans), triangles so large that the grown
// box covers more than RING_SCAN_CELLS cells.
#pragma once
#include "oa_tri.hpp"

namespace oa {

constexpr int RING_SCAN_CELLS = 150;      // cells the build looks at per triangle before it gives the triangle up

#if defined(__HIPCC__)

enum { RING_STAT_TRIS, RING_STAT_NEVER, RING_STAT_RECORDS, RING_STAT_TESTS, RING_STAT_NEIGHBOURS, RING_STAT_CAPPED, RING_STAT_N };

struct TriAxes {                          // the seven axes of T (unit, double) and T's support along them
    double u[7][3];                       // 0: normal; 1..3: outward edge normals of ab, bc, ca; 4..6: centroid -> a, b, c
    double lo0, hi0;                      // normal axis: [min, max] of n . corner
    double h[7];                          // axes 1..6: max over T's corners of u . x
};

__device__ __forceinline__ double dot3d(const double *a, const double *b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }

// false: degenerate (no normal), or non-finite
__device__ __forceinline__ bool tri_axes(const double A[3], const double B[3], const double C[3], TriAxes &ax)
{
    double ab[3], bc[3], ca[3], g[3];
    for (int k = 0; k < 3; ++k) { ab[k] = B[k] - A[k]; bc[k] = C[k] - B[k]; ca[k] = A[k] - C[k]; g[k] = (A[k] + B[k] + C[k]) * (1.0 / 3.0); }
    double N[3] = { ab[1] * (-ca[2]) - ab[2] * (-ca[1]), ab[2] * (-ca[0]) - ab[0] * (-ca[2]), ab[0] * (-ca[1]) - ab[1] * (-ca[0]) };   // ab x ac
    const double n2 = dot3d(N, N);
    const double e2 = fmax(fmax(dot3d(ab, ab), dot3d(bc, bc)), dot3d(ca, ca));
    // (needles: a normal from a cancelling cross product points anywhere -- it would still be a unit vector, i.e. a valid axis,
    //  but the edge normals built on it would not separate anything; such triangles are left to the rings of cells)
    if (!(n2 > 1e-12 * e2 * e2) || !(n2 < 1e300)) return false;
    const double inv = 1.0 / sqrt(n2);
    for (int k = 0; k < 3; ++k) ax.u[0][k] = N[k] * inv;
    const double *edge[3] = { ab, bc, ca };
    const double *opp[3] = { C, A, B };                              // the corner opposite each edge
    const double *org[3] = { A, B, C };                              // a point on each edge
    for (int e = 0; e < 3; ++e) {
        const double *d = edge[e];
        double m[3] = { d[1] * ax.u[0][2] - d[2] * ax.u[0][1], d[2] * ax.u[0][0] - d[0] * ax.u[0][2], d[0] * ax.u[0][1] - d[1] * ax.u[0][0] };   // d x n: in the plane, normal to the edge
        const double m2 = dot3d(m, m);
        if (!(m2 > 0.0)) return false;
        const double w[3] = { opp[e][0] - org[e][0], opp[e][1] - org[e][1], opp[e][2] - org[e][2] };
        const double s = (dot3d(m, w) > 0.0 ? -1.0 : 1.0) / sqrt(m2);   // away from the opposite corner
        for (int k = 0; k < 3; ++k) ax.u[1 + e][k] = m[k] * s;
    }
    const double *crn[3] = { A, B, C };
    for (int v = 0; v < 3; ++v) {
        double d[3] = { crn[v][0] - g[0], crn[v][1] - g[1], crn[v][2] - g[2] };
        const double d2 = dot3d(d, d);
        if (!(d2 > 0.0)) return false;
        const double s = 1.0 / sqrt(d2);
        for (int k = 0; k < 3; ++k) ax.u[4 + v][k] = d[k] * s;
    }
    const double na = dot3d(ax.u[0], A), nb = dot3d(ax.u[0], B), nc = dot3d(ax.u[0], C);
    ax.lo0 = fmin(fmin(na, nb), nc); ax.hi0 = fmax(fmax(na, nb), nc);
    ax.h[0] = 0.0;
    for (int a = 1; a < 7; ++a) ax.h[a] = fmax(fmax(dot3d(ax.u[a], A), dot3d(ax.u[a], B)), dot3d(ax.u[a], C));
    return ax.lo0 == ax.lo0 && ax.hi0 == ax.hi0;
}

// lower bound of dist(T, {y : |y - c| <= r}) from T's axes (>= 0)
__device__ __forceinline__ double tri_axes_bound_ball(const TriAxes &ax, const double c[3], double r)
{
    const double pn = dot3d(ax.u[0], c);
    double g = fmax(pn - ax.hi0, ax.lo0 - pn);
#pragma unroll
    for (int a = 1; a < 7; ++a) g = fmax(g, dot3d(ax.u[a], c) - ax.h[a]);
    return fmax(g - r, 0.0);
}

// lower bound of dist(T, T') from T's axes, T' = (P, Q, R)  (>= 0; NaN corners give NaN)
__device__ __forceinline__ double tri_axes_bound_tri(const TriAxes &ax, const double P[3], const double Q[3], const double R[3])
{
    const double p0 = dot3d(ax.u[0], P), q0 = dot3d(ax.u[0], Q), r0 = dot3d(ax.u[0], R);
    double g = fmax(fmin(fmin(p0, q0), r0) - ax.hi0, ax.lo0 - fmax(fmax(p0, q0), r0));
#pragma unroll
    for (int a = 1; a < 7; ++a) {
        const double m = fmin(fmin(dot3d(ax.u[a], P), dot3d(ax.u[a], Q)), dot3d(ax.u[a], R));
        g = fmax(g, m - ax.h[a]);
    }
    // a corner that is not finite: no bound (fmax / fmin drop NaN operands -- and a triangle with an infinite corner can still
    // be somebody's nearest through its finite ones)
    const double chk = (fabs(P[0]) + fabs(P[1]) + fabs(P[2])) + (fabs(Q[0]) + fabs(Q[1]) + fabs(Q[2])) + (fabs(R[0]) + fabs(R[1]) + fabs(R[2]));
    if (!(chk < INFINITY) || g != g) return NAN;
    return fmax(g, 0.0);
}

// One thread per triangle, two phases per batch of cell rows, so that divergence does not multiply the expensive part (the first
// version walked straight through: with 64 lanes at different records, nearly every step of the wave took the slow path of
// SOME lane -- three dependent loads and the seven-axis test in double -- and a build cost 6.4 ms at 1.96M triangles):
//   1  the records of the cells around T, four per trip with all their loads in flight: skip T itself and every record that
//      is not the FIRST of its triangle inside this scan (the flags k_tri_grid_bin leaves in the index word say whether the
//      record's cell is the triangle's lowest along x / y / z: first <=> per axis "lowest cell of the triangle, or lowest cell
//      of the scan"), then the ball-against-ball test in float -- survivors go on the lane's queue in LDS;
//   2  the queue: the triangle itself (three loads), ball against T's axes, the seven-axis bound; neighbours are appended to
//      the list, the others lower m.  All lanes are here together.
// tri9 is read (corners) and written (only .y of the third float4 of the thread's OWN triangle: no other thread reads that
// lane).  cap: the clearance beyond which nobody asks (a fraction of the cell edge).
constexpr int RING_QUEUE = 24;            // queue entries per lane (x 256 lanes x 4 B of LDS)
template <bool STATS>
__global__ __launch_bounds__(256) void k_tri_ring_build(float4 *__restrict__ tri9, int n_tris, GridParams gp,
                                                        const int *__restrict__ cell_start, const float4 *__restrict__ cell_rec,
                                                        double cap, int *__restrict__ ring, unsigned long long *__restrict__ stats)
{
    __shared__ int queue[RING_QUEUE][256];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = t < n_tris;
    int *my = ring + (size_t)TRI_RING_STRIDE * (size_t)(have ? t : 0);
    float accept = 0.f;
    int n_ring = 0, n_q = 0;
    unsigned long long n_rec = 0, n_test = 0;
    bool capped = false;
    double A[3] = { 0, 0, 0 }, B[3] = { 0, 0, 0 }, C[3] = { 0, 0, 0 };
    if (have) {
        const float4 u = tri9[3ll * t], v = tri9[3ll * t + 1], w = tri9[3ll * t + 2];
        A[0] = u.x; A[1] = u.y; A[2] = u.z; B[0] = u.w; B[1] = v.x; B[2] = v.y; C[0] = v.z; C[1] = v.w; C[2] = w.x;
    }
    TriAxes ax;
    bool ok = have && tri_axes(A, B, C, ax);
    int lo[3] = { 0, 0, 0 }, hi[3] = { -1, -1, -1 };
    double g[3] = { 0, 0, 0 }, rad = 0.0;
    if (ok) {
        long long cells = 1;
        for (int a = 0; a < 3; ++a) {
            const double mn = fmin(fmin(A[a], B[a]), C[a]) - cap - gp.slack, mx = fmax(fmax(A[a], B[a]), C[a]) + cap + gp.slack;
            lo[a] = grid_cell_coord(mn, gp.lo[a], gp.inv_h, gp.n[a]);
            hi[a] = grid_cell_coord(mx, gp.lo[a], gp.inv_h, gp.n[a]);
            cells *= (long long)(hi[a] - lo[a] + 1);
            g[a] = (A[a] + B[a] + C[a]) * (1.0 / 3.0);
        }
        if (cells > RING_SCAN_CELLS) ok = false;
        for (int v = 0; v < 3; ++v) {
            const double *P = v == 0 ? A : (v == 1 ? B : C);
            const double d2 = (P[0] - g[0]) * (P[0] - g[0]) + (P[1] - g[1]) * (P[1] - g[1]) + (P[2] - g[2]) * (P[2] - g[2]);
            rad = fmax(rad, d2);
        }
        rad = sqrt(rad) * (1.0 + 1e-12);                            // T lies within rad of its centroid
    }
    if (!ok) { hi[0] = lo[0] - 1; hi[1] = lo[1] - 1; hi[2] = lo[2] - 1; }     // (an empty scan: the lane stays for the wave's loops)
    const double tau = 1e-3 * cap;                                  // nearer than this: a neighbour
    double m = cap;                                                 // smallest bound over the non-neighbours so far
    // float images for phase 1: the ball test  |c' - g| > r' + rad + cap  =>  farther than cap (margins: 1e-5 relative covers the
    // float roundings of both sides)
    const float gx = (float)g[0], gy = (float)g[1], gz = (float)g[2];
    const float reach0 = (float)((rad + cap) * (1.0 + 1e-5) + 4e-7 * gp.scale);

    // phase 2 (the whole wave): the lane's queued triangles against T
    auto drain = [&]() {
        int n_max = n_q;
        for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, __shfl_xor(n_max, o, 64));
        for (int k = 0; k < n_max; ++k) {
            if (k < n_q && ok) {
                const uint32_t o = (uint32_t)queue[k][threadIdx.x];
                double P[3], Q[3], R[3];
                {
                    const float4 u = tri9[3ll * o], v = tri9[3ll * o + 1], w = tri9[3ll * o + 2];
                    P[0] = u.x; P[1] = u.y; P[2] = u.z; Q[0] = u.w; Q[1] = v.x; Q[2] = v.y; R[0] = v.z; R[1] = v.w; R[2] = w.x;
                }
                if (STATS) ++n_test;
                const double lb = tri_axes_bound_tri(ax, P, Q, R);
                if (!(lb > tau)) {                                   // (NaN = no bound: a neighbour, evaluated with the rest)
                    if (n_ring >= TRI_RING_MAX) ok = false;          // a fan: left to the rings of cells
                    else my[n_ring++] = (int)o;
                } else if (lb < m) m = lb;
            }
        }
        n_q = 0;
    };

    int z = lo[2], y = lo[1];
    bool scanning = hi[2] >= lo[2] && hi[1] >= lo[1] && hi[0] >= lo[0];
    while (__any(scanning)) {
        // this lane's next row of cells: the records of cells lo[0] .. hi[0] are one range
        int ja = 0, jm = 0, jb = 0;
        bool fy_lo = false, fz_lo = false;
        if (scanning) {
            const int row = (z * gp.n[1] + y) * gp.n[0];
            ja = cell_start[row + lo[0]]; jm = cell_start[row + lo[0] + 1]; jb = cell_start[row + hi[0] + 1];
            fy_lo = y == lo[1]; fz_lo = z == lo[2];
            if (++y > hi[1]) { y = lo[1]; if (++z > hi[2]) scanning = false; }
        }
        int j = ja;
        while (__any(j < jb)) {
            // (no branch around the loads: a lane that is through reads record 0 and drops it)
            float4 r0[4];
            uint32_t w1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = j + u < jb ? j + u : 0;
                r0[u] = tri_ld_rec(cell_rec, e, 0);
                w1[u] = __float_as_uint(tri_ld_rec(cell_rec, e, 1).w);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bool keep = j + u < jb && ok;
                if (STATS && keep) ++n_rec;
                const uint32_t o = w1[u] & TRI_REC_INDEX_MASK;
                // first record of its triangle in this scan?
                const bool first = ((w1[u] & TRI_REC_FLAG_X) || j + u < jm) && ((w1[u] & TRI_REC_FLAG_Y) || fy_lo) && ((w1[u] & TRI_REC_FLAG_Z) || fz_lo);
                keep = keep && first && o != (uint32_t)t;
                if (keep) {
                    const float dx = r0[u].x - gx, dy = r0[u].y - gy, dz = r0[u].z - gz;
                    const float rs = r0[u].w + reach0;
                    const float dd = dx * dx + dy * dy + dz * dz;
                    if (dd > rs * rs * 1.00001f && dd < INFINITY) keep = false;             // (a record that is not finite: kept -- no bound)
                }
                if (keep) queue[n_q++][threadIdx.x] = (int)o;
            }
            if (j < jb) j += 4;
            if (__any(n_q > RING_QUEUE - 4)) drain();
        }
    }
    drain();
    if (ok) {
        capped = !(m < cap);
        // A = C / 2 with the margins of the header; 0 when nothing is left of it
        const double a = 0.5 * (m - gp.slack) * (1.0 - 1e-4) - 1e-12 * gp.scale;
        float af = a > 0.0 && a < 3.0e38 ? (float)a : 0.f;
        if ((double)af > a) af = nextafterf(af, 0.f);
        if (!(af >= 1e-30f)) af = 0.f;
        accept = af;
    }
    if (!have) return;
    if (!ok) n_ring = 0;
    for (int k = n_ring; k < TRI_RING_MAX; ++k) my[k] = -1;
    my[TRI_RING_MAX] = n_ring;
    ((float *)&tri9[3ll * t + 2])[1] = accept;
    if (STATS && stats) {
        // per-workgroup totals would be kinder; this is an instrumented build only
        atomicAdd(&stats[RING_STAT_TRIS], 1ull);
        if (!(accept > 0.f)) atomicAdd(&stats[RING_STAT_NEVER], 1ull);
        atomicAdd(&stats[RING_STAT_RECORDS], n_rec);
        atomicAdd(&stats[RING_STAT_TESTS], n_test);
        atomicAdd(&stats[RING_STAT_NEIGHBOURS], (unsigned long long)n_ring);
        if (capped) atomicAdd(&stats[RING_STAT_CAPPED], 1ull);
    }
}

// diagnostic (oa_get_stat OA_STAT_TRI_RING_ACCEPTS): how many queries the neighbour lists would settle at the current pose with
// the current seeds -- the test of k_tri_search_grid's prologue, nothing written but the count
#if !defined(OA_FAMILY_TU) && defined(OA_EXPERIMENTS)      // experiment: only in liboa_icp_exp.so's host translation unit
__global__ __launch_bounds__(256) void k_tri_ring_count(const DevState *__restrict__ st, const float4 *__restrict__ src4, int ns, double scale,
                                                        const float4 *__restrict__ tri9, const int *__restrict__ prev,
                                                        unsigned long long *__restrict__ out, unsigned long long *__restrict__ out_why)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool accepted = false;
    int why = 1;
    if (i < ns) {
        const float4 p4 = src4[i];
        float pf[3];
        co_find(st, p4.x, p4.y, p4.z, pf[0], pf[1], pf[2]);
        const int s = prev[i];
        if (s >= 0) {
            float a[3], b[3], c[3], r[3];
            const float4 u = tri9[3ll * s], v = tri9[3ll * s + 1], w = tri9[3ll * s + 2];
            a[0] = u.x; a[1] = u.y; a[2] = u.z; b[0] = u.w; b[1] = v.x; b[2] = v.y; c[0] = v.z; c[1] = v.w; c[2] = w.x;
            closest_on_tri(pf, a, b, c, r);
            const float d = tri_dist2(pf, r);
            const float scalef = (float)scale * 1.000001f;
            const float deltaf = (fabsf(pf[0]) + fabsf(pf[1]) + fabsf(pf[2]) + scalef) * 3.8148e-6f + scalef * 1.1e-10f;
            accepted = d < INFINITY && tri_ring_accepts(d, w.y, deltaf) && tri_seed_certified(pf, a, b, c);
            if (!accepted) {
                if (!(w.y > 0.f)) why = 2;
                else if (d < INFINITY && tri_ring_accepts(d, w.y, deltaf)) why = 3;
                else {
                    const float need = sqrtf(d) + 2.5f * deltaf;
                    why = need <= w.y ? 4 : (need <= 2.f * w.y ? 5 : (need <= 4.f * w.y ? 6 : (need <= 8.f * w.y ? 7 : 8)));
                }
            }
        }
    }
    const unsigned long long m = __ballot(accepted);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
    // OA_DEBUG: why not -- out[1]: no seed; [2]: the seed never accepts (radius 0); [3]: not certified; [4 + k]: sqrt(d2) + 2.5 delta is
    // within 2^k of the radius (k = 0 .. 3: up to 1x / 2x / 4x / 8x), [8]: farther
    if (out_why && i < ns && !accepted) atomicAdd(out_why + why, 1ull);
}
#endif  // !OA_FAMILY_TU

#endif  // __HIPCC__
}  // namespace oa
