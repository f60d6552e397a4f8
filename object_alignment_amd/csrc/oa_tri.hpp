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

// sum of triangle bounding-box diagonals (for the cell size) -- double atomics are fine here (one-time, not a result)
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
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0 && d > 0.0) atomicAdd(out, d);
}

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

// bounding sphere of a triangle for the cell lists: centre = centre of its bounding box, radius rounded UP, so that
// every point of the triangle is within r of the centre (the search subtracts r from the distance to the centre)
__device__ __forceinline__ float4 tri_sphere(const float *a, const float *b, const float *c)
{
    float ctr[3];
    for (int k = 0; k < 3; ++k)
        ctr[k] = 0.5f * fminf(fminf(a[k], b[k]), c[k]) + 0.5f * fmaxf(fmaxf(a[k], b[k]), c[k]);
    double r2 = 0.0;
    const float *v[3] = { a, b, c };
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) { const double d = (double)v[i][k] - (double)ctr[k]; s += d * d; }
        r2 = s > r2 ? s : r2;
    }
    const double r = sqrt(r2) * (1.0 + 1e-6) + 1e-37;
    return make_float4(ctr[0], ctr[1], ctr[2], r < 3.0e38 ? (float)r : INFINITY);   // NaN stays NaN: never skipped, never selected
}

// pass 0: counts[cell] += 1 for every cell overlapped; pass 1: write the triangle id (and its bounding sphere, so the
// search can discard most candidates from one contiguous 16-byte record) at cell_start[cell] + cursor++
template <bool FILL>
__global__ void k_tri_grid_bin(const float4 *__restrict__ tri9, int n_tris, GridParams gp, int *__restrict__ counts,
                               const int *__restrict__ cell_start, int *__restrict__ cell_tris,
                               float4 *__restrict__ cell_sph, unsigned long long *__restrict__ total)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tris) return;
    int lo[3], hi[3];
    bool ok;
    tri_cell_range(tri9, t, gp, lo, hi, ok);
    if (!ok) return;
    float4 sph = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FILL) {
        float a[3], b[3], c[3];
        load_tri(tri9, t, a, b, c);
        sph = tri_sphere(a, b, c);
    }
    unsigned long long n = 0;
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const int cidx = (z * gp.n[1] + y) * gp.n[0] + x;
                if (FILL) {
                    const int pos = cell_start[cidx] + atomicAdd(&counts[cidx], 1);
                    cell_tris[pos] = t;
                    cell_sph[pos] = sph;
                } else atomicAdd(&counts[cidx], 1);
                ++n;
            }
    if (!FILL && total) atomicAdd(total, n);
}

__device__ __forceinline__ void tri_eval(const float *p, const float4 *__restrict__ tri9, uint32_t t, float &best, uint32_t &bidx)
{
    float a[3], b[3], c[3], r[3];
    load_tri(tri9, t, a, b, c);
    closest_on_tri(p, a, b, c, r);
    const float d = tri_dist2(p, r);
    if (d < best || (d == best && t < bidx && d < INFINITY)) { best = d; bidx = t; }
}

// Squared-gap threshold above which a triangle's bounding box proves it can neither beat nor tie `best`: the float32
// closest-point evaluation is >= (D - delta)^2 (1 - 1e-5) for real distance D, so a box at squared gap
// > (delta + sqrt((best + 1e-30) / (1 - 1e-5)))^2 is out.  The extra 3e-6 covers the float rounding of the gap (6u) and
// of the threshold itself.
__device__ __forceinline__ float tri_skip_threshold(float best, double delta)
{
    if (!(best < INFINITY)) return INFINITY;
    const double s = delta + sqrt(((double)best + 1e-30) / (1.0 - 1e-5));
    return (float)(s * s * (1.0 + 3e-6));
}

// sqrt of the threshold, rounded up: how far from the query a triangle may be and still matter (sphere test)
__device__ __forceinline__ float tri_reach(float thr)
{
    if (!(thr < INFINITY)) return INFINITY;
    const double r = sqrt((double)thr) * (1.0 + 1e-6) + 1e-37;
    return r < 3.0e38 ? (float)r : INFINITY;
}

struct TriSearchState {
    float best; uint32_t bidx;
    float lim, thr, reach;    // lim = min(best, search radius^2); thr = squared-gap threshold; reach = sqrt(thr), rounded up
    double reach2;            // (delta + sqrt((lim + 1e-30) / (1 - 1e-5)))^2: rows / rings whose squared gap exceeds it are out
};

// everything derived from `lim` (called when the best improves: rare)
__device__ __forceinline__ void tri_state_refresh(TriSearchState &s, double delta)
{
    s.thr = tri_skip_threshold(s.lim, delta);
    s.reach = tri_reach(s.thr);
    if (s.lim < INFINITY) {
        const double r = (delta + sqrt(((double)s.lim + 1e-30) / (1.0 - 1e-5))) * (1.0 + 1e-9);
        s.reach2 = r * r;
    } else s.reach2 = INFINITY;
}

// Cell-list candidates go through two phases so that divergence does not multiply the expensive part.  Phase 1
// (tri_candidate): sphere test on the contiguous 16-byte record; survivors' triangle ids are pushed on a per-thread
// queue in LDS.  Phase 2 (tri_queue_flush): every lane evaluates ITS k-th survivor in the same trip, so a wave pays
// max-over-lanes(survivors) closest-point evaluations instead of one per candidate slot in which any lane survived
// (measured with SQ_INSTS_VALU: 18.5k -> 9.5k instructions per wave, profiles/r01g_surface_1M_pmc_summary.txt).
constexpr int TRI_QUEUE = 16;

__device__ __forceinline__ void tri_candidate(const float *p, const float4 sph, int tid, const TriSearchState &s,
                                              int (*queue)[256], int &nq)
{
    const float dx = sph.x - p[0], dy = sph.y - p[1], dz = sph.z - p[2];
    const float D2 = dx * dx + dy * dy + dz * dz;
    const float rs = sph.w + s.reach;
    if (D2 > rs * rs * 1.000003f) return;                          // farther than radius + reach: cannot beat or tie
    queue[nq][threadIdx.x] = tid;                                  // the caller keeps nq <= TRI_QUEUE - 4 before a trip
    ++nq;
}

// Phase 2.  The triangle of survivor k + 1 is fetched before survivor k is evaluated: a thread's time is a chain of
// memory round trips, and the ~300-instruction evaluation hides the next fetch instead of following it.
__device__ __forceinline__ void tri_queue_flush(const float *p, const float4 *__restrict__ tri9, TriSearchState &s,
                                                int (*queue)[256], int &nq, double delta, float cutf)
{
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f), v = u, w = u;
    uint32_t t = IDX_NONE;
    if (nq > 0) { t = (uint32_t)queue[0][threadIdx.x]; u = tri9[3ll * t]; v = tri9[3ll * t + 1]; w = tri9[3ll * t + 2]; }
    for (int k = 0; __any(k < nq); ++k) {
        float4 nu = u, nv = v, nw = w;
        uint32_t nt = IDX_NONE;
        if (k + 1 < nq) { nt = (uint32_t)queue[k + 1][threadIdx.x]; nu = tri9[3ll * nt]; nv = tri9[3ll * nt + 1]; nw = tri9[3ll * nt + 2]; }
        if (k < nq && t != s.bidx) {                               // (already the best: listed in several cells)
            const float a[3] = { u.x, u.y, u.z }, b[3] = { u.w, v.x, v.y }, c[3] = { v.z, v.w, w.x };
            float lb = 0.f;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float lo = fminf(fminf(a[m], b[m]), c[m]), hi = fmaxf(fmaxf(a[m], b[m]), c[m]);
                const float g = fmaxf(fmaxf(lo - p[m], p[m] - hi), 0.f);
                lb += g * g;
            }
            if (!(lb > s.thr)) {
                float r[3];
                closest_on_tri(p, a, b, c, r);
                const float d = tri_dist2(p, r);
                if (d < s.best || (d == s.best && t < s.bidx && d < INFINITY)) {
                    const bool closer = d < s.best;
                    s.best = d; s.bidx = t;
                    if (closer) { s.lim = fminf(s.best, cutf); tri_state_refresh(s, delta); }
                }
            }
        }
        t = nt; u = nu; v = nv; w = nw;
    }
    nq = 0;
}

// L = 1, 2 or 4 lanes per query, as in k_nn_search_grid: the rows of a ring are dealt out to the lanes, every lane runs
// both phases on its rows with its own state, and the lanes merge (d2, index) after every batch of rows.
template <int L>
__global__ __launch_bounds__(256, 4) void k_tri_search_grid(const DevState *__restrict__ st,
                                                         const float4 *__restrict__ src4, int ns, GridParams gp,
                                                         const int *__restrict__ cell_start,
                                                         const int *__restrict__ cell_tris,
                                                         const float4 *__restrict__ cell_sph,
                                                         const float4 *__restrict__ tri9,
                                                         const int *__restrict__ prev,
                                                         unsigned long long *__restrict__ keys,
                                                         int *__restrict__ todo_list, int *__restrict__ todo_count, int turn)
{
    constexpr int RPL = (9 + L - 1) / L;                            // rows per lane and batch
    constexpr int BATCH = RPL * L;
    if (st->halt) return;
    if (turn >= 0 && (st->tree_turn != 0) != (turn != 0)) return;  // not this kernel's turn (DevState::tree_turn)
    __shared__ int queue[TRI_QUEUE][256];
    int nq = 0;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gt / L, sub = gt % L;                             // the L lanes of a query are neighbours in a wave
    if (i >= ns) return;
    const float4 p4 = src4[i];
    float wx, wy, wz, pf[3];
    m4_mul_v3(st->mx1, p4.x, p4.y, p4.z, wx, wy, wz);
    m4_mul_v3(st->imx2, wx, wy, wz, pf[0], pf[1], pf[2]);          // co_find (general.py:287)

    float best = INFINITY;
    uint32_t bidx = IDX_NONE;
    const int s = prev ? prev[i] : -1;
    if (s >= 0) tri_eval(pf, tri9, (uint32_t)s, best, bidx);

    const double p[3] = { (double)pf[0], (double)pf[1], (double)pf[2] };
    double pc[3], off2 = 0.0, pabs = 0.0;
    int c[3];
    bool finite = true;
    for (int a = 0; a < 3; ++a) {
        if (!(fabs(p[a]) < INFINITY)) finite = false;
        pabs += fabs(p[a]);
        pc[a] = p[a] < gp.lo[a] ? gp.lo[a] : (p[a] > gp.hi[a] ? gp.hi[a] : p[a]);
        const double d = p[a] - pc[a];
        off2 += d * d;
        c[a] = grid_cell_coord(pc[a], gp.lo[a], gp.inv_h, gp.n[a]);
    }
    // float32 closest-point evaluation can undershoot the real distance by at most delta
    const double delta = 64.0 * 5.9604644775390625e-08 * (gp.scale + pabs) + gp.slack;
    // `lim`: the best so far or the search radius (search_cutoff2), whichever is smaller -- see k_nn_search_grid
    const float cutf = search_cutoff2(st, pf[0], pf[1], pf[2]);
    TriSearchState S;
    S.best = best; S.bidx = bidx;
    S.lim = fminf(best, cutf);
    tri_state_refresh(S, delta);
    bool settled = false, over = false;
    // candidates this query may look at (see GridParams; split between its lanes).  While the pose still moves by a good part of a cell per
    // iteration (first iteration, or last iteration's translation + rotation x object size above h / 4) the seeds are
    // stale and most queries need the second ring: handing them all to the tree costs more than letting the grid
    // look at twice as many candidates.
    int budget = gp.budget;
    {
        const int last = (st->n + 4) % 5;
        const double moved = st->use_target && st->n > 0 ? (st->ring_t[last] + st->ring_r[last] * gp.scale) * st->local_per_world : 0.0;
        if (st->n == 0 || moved > 0.25 * gp.h) budget *= 2;
        if (L > 1) budget = budget / L + 8;
    }
    const int r_start = (S.bidx != IDX_NONE && gp.seeded_start) ? 1 : 0;   // as in k_nn_search_grid
    if (finite) {
        for (int r = r_start; r <= gp.r_max && !settled && !over; ++r) {
            const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, gp.n[0] - 1);
            // rows of the ring nine at a time: cell ranges first (independent loads), then the candidates -- as in
            // k_nn_search_grid
            const int side = 2 * r + 1, n_rows = side * side;
            const unsigned div_mul = 65536u / (unsigned)side + 1u;
            for (int b0 = 0; b0 < n_rows && !over; b0 += BATCH) {
                int ja[RPL], jb[RPL], jc[RPL], jd[RPL];
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    ja[k] = jb[k] = jc[k] = jd[k] = 0;
                    const int kk = b0 + sub + L * k;
                    if (kk >= n_rows) continue;
                    const int qz = (int)(((unsigned)kk * div_mul) >> 16);
                    const int dzi = qz - r, dyi = kk - qz * side - r;
                    const int z = c[2] + dzi, y = c[1] + dyi;
                    if (z < 0 || z >= gp.n[2] || y < 0 || y >= gp.n[1]) continue;
                    const double dz = grid_axis_gap(pc[2], gp.lo[2], gp.h, z, gp.slack);
                    const double dy = grid_axis_gap(pc[1], gp.lo[1], gp.h, y, gp.slack);
                    const double row2 = off2 + dz * dz + dy * dy;
                    // (sqrt(row2) - delta)^2 (1 - 1e-5) - 1e-30 > lim  <=>  row2 > reach2: nothing in this row can matter
                    if (row2 * (1.0 - 1e-9) > S.reach2) continue;
                    // cells of the row whose slab along x can still hold a triangle within reach: |x - pc.x| <= w
                    int xa = x0, xb = x1;
                    if (S.reach2 < 1e300) {
                        const double w2 = S.reach2 - row2 * (1.0 - 1e-9);
                        const double w = (double)grid_sqrt_up((float)(w2 > 0.0 ? w2 * (1.0 + 1e-6) : 0.0)) + gp.slack;
                        xa = max(xa, grid_cell_coord(pc[0] - w, gp.lo[0], gp.inv_h, gp.n[0]));
                        xb = min(xb, grid_cell_coord(pc[0] + w, gp.lo[0], gp.inv_h, gp.n[0]));
                    }
                    const int row = (z * gp.n[1] + y) * gp.n[0];
                    const bool shell_row = (r == r_start) || dzi == -r || dzi == r || dyi == -r || dyi == r;
                    if (shell_row) {
                        if (xa <= xb) { ja[k] = cell_start[row + xa]; jb[k] = cell_start[row + xb + 1]; }
                    } else {
                        const int xl = c[0] - r, xr = c[0] + r;
                        if (xl >= xa && xl <= xb) { ja[k] = cell_start[row + xl]; jb[k] = cell_start[row + xl + 1]; }
                        if (xr >= xa && xr <= xb) { jc[k] = cell_start[row + xr]; jd[k] = cell_start[row + xr + 1]; }
                    }
                }
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
#pragma unroll
                    for (int sg = 0; sg < 2; ++sg) {
                        const int j0 = sg ? jc[k] : ja[k], j1 = sg ? jd[k] : jb[k];
                        if (j1 <= j0) continue;
                        budget -= j1 - j0;
                        if (budget < 0) break;                       // crowded cells: one wave of the tree search is faster
                        const int last = j1 - 1;
                        for (int j = j0; j < j1; j += 4) {
                            if (__any(nq > TRI_QUEUE - 4)) tri_queue_flush(pf, tri9, S, queue, nq, delta, cutf);
                            // the records and (whether or not they pass) their triangle ids: eight independent loads
                            const int e1 = min(j + 1, last), e2 = min(j + 2, last), e3 = min(j + 3, last);
                            const float4 s0 = cell_sph[j], s1 = cell_sph[e1], s2 = cell_sph[e2], s3 = cell_sph[e3];
                            const int t0 = cell_tris[j], t1 = cell_tris[e1], t2 = cell_tris[e2], t3 = cell_tris[e3];
                            tri_candidate(pf, s0, t0, S, queue, nq);
                            if (j + 1 < j1) tri_candidate(pf, s1, t1, S, queue, nq);
                            if (j + 2 < j1) tri_candidate(pf, s2, t2, S, queue, nq);
                            if (j + 3 < j1) tri_candidate(pf, s3, t3, S, queue, nq);
                        }
                    }
                }
                tri_queue_flush(pf, tri9, S, queue, nq, delta, cutf);   // a better best prunes the next batch of rows
                over = budget < 0;
                if (L > 1) {                                         // the lanes of the query agree on the best so far
                    bool changed = false;
#pragma unroll
                    for (int o = 1; o < L; o <<= 1) {
                        const float ob = __shfl_xor(S.best, o, 64);
                        const uint32_t oi = (uint32_t)__shfl_xor((int)S.bidx, o, 64);
                        if (ob < S.best) { S.best = ob; S.bidx = oi; changed = true; }
                        else if (ob == S.best && oi < S.bidx) S.bidx = oi;
                        over = (__shfl_xor((int)over, o, 64) != 0) || over;
                    }
                    if (changed) { S.lim = fminf(S.best, cutf); tri_state_refresh(S, delta); }
                }
            }
            if (over) break;
            double m = INFINITY;
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) { const double f = pc[a] - (gp.lo[a] + (double)(c[a] - r) * gp.h); m = f < m ? f : m; }
                if (c[a] + r + 1 < gp.n[a]) { const double f = (gp.lo[a] + (double)(c[a] + r + 1) * gp.h) - pc[a]; m = f < m ? f : m; }
            }
            if (!(m < INFINITY)) settled = true;
            else {
                m -= gp.slack;
                m = m > 0.0 ? m : 0.0;
                if ((off2 + m * m) * (1.0 - 1e-9) > S.reach2) settled = true;   // same test as for a row
            }
        }
    }
    if (sub != 0) return;
    keys[i] = ((unsigned long long)__float_as_uint(S.best) << 32) | S.bidx;
    if (!settled) todo_list[atomicAdd(todo_count, 1)] = i;
}

// brute force over all triangles for every source point (OA_SEARCH_BRUTE; the oracle's oo_nn_tri_brute on the device)
__global__ __launch_bounds__(256) void k_tri_search_all(const DevState *__restrict__ st,
                                                        const float4 *__restrict__ src4, int ns,
                                                        const float4 *__restrict__ tri9, int n_tris,
                                                        const int *__restrict__ prev,
                                                        unsigned long long *__restrict__ keys)
{
    if (st->halt) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const float4 p4 = src4[i];
        float wx, wy, wz, pf[3];
        m4_mul_v3(st->mx1, p4.x, p4.y, p4.z, wx, wy, wz);
        m4_mul_v3(st->imx2, wx, wy, wz, pf[0], pf[1], pf[2]);
        float best = INFINITY;
        uint32_t bidx = IDX_NONE;
        if (prev && prev[i] >= 0) tri_eval(pf, tri9, (uint32_t)prev[i], best, bidx);
        for (int t = 0; t < n_tris; ++t) tri_eval(pf, tri9, (uint32_t)t, best, bidx);
        keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
    }
}

#endif  // __HIPCC__
}  // namespace oa
