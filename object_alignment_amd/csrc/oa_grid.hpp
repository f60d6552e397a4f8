// oa_grid.hpp -- exact nearest-vertex search through a uniform grid.
//
// Same answers as k_nn_search (bit-identical (d2, index) per source point): candidates are evaluated with the same
// fp32 difference-form metric on the ORIGINAL coordinates and merged lexicographically on (d2, original index); the
// grid only decides which vertices need not be looked at, through a conservative bound.
//
// Build (once per target): bounding box -> cell ids -> histogram (atomics) -> exclusive scan -> scatter into
// `sorted` (float4 {x, y, z, bits(original index)}), cells laid out x-fastest so a run of cells along x is one
// contiguous range of `sorted`.
//
// Query (one thread per source point): project the query onto the grid's box (pc), search the cube of cells within
// Chebyshev radius r = 0, 1, 2, ... around pc's cell.  Every vertex not yet seen lies outside that cube, hence at
// real distance >= sqrt(|p - pc|^2 + m^2), m = distance from pc to the nearest cube face that is interior to the
// grid.  The metric satisfies d2 >= D (1 - 5.01 u); the loop stops when (|p-pc|^2 + m^2)(1 - 1e-5) > lim, i.e. no
// unseen vertex can beat OR tie the current best (or lie within the search radius, DevState::cut_a).  Queries not
// settled within `r_max` rings, or whose cells hold more than `budget` candidates, are appended to a list that the
// tree search (oa_bvh.hpp: k_bvh_search) finishes, so the result is exact for any input.
#pragma once
#include "oa_kernels.hpp"

namespace oa {

struct GridParams {
    double lo[3], hi[3];     // bounding box of the target
    double h, inv_h;         // cell edge
    int    n[3];             // cells per axis
    int    r_max;            // rings searched before handing the query to the brute-force kernel
    double slack;            // absolute slack subtracted from face distances (1e-10 x largest |coordinate|)
    double scale;            // largest |coordinate| of the box
    int    seeded_start;     // 1: a seeded query starts with the 3 x 3 x 3 block instead of its own cell (OA_GRID_SEEDED_START)
    int    budget_moving;    // ... while the pose still moves by a good part of a cell per iteration, or in the first search of a loop
    int    budget;           // candidates one thread may look at before it hands the query to the tree search
                             // (crowded cells -- clusters, fans of thin triangles -- would otherwise stall its wave)
    // float images for the per-row arithmetic of the searches (GridQuery): cell edge, its inverse, and a slack that
    // covers `slack` above plus every float rounding between the double cell frame and a gap (< 2e-6 h), rounded up
    float  hf, inv_hf, slackf;
    float  eps_plane;        // triangle grid: how far a triangle's corners may lie from the plane of its record (oa_tri.hpp)
    int    drop_over;        // triangle grid: a query whose listed cells exceed its budget scans none of them (OA_TRI_DROP_OVER)
    double moving_h;         // the pose "still moves" (budget_moving applies) while last step's translation + rotation x size exceeds this
    double xcd_moving_h;     // triangle grid: ... while last step's motion exceeds THIS (a multiple of moving_h: far queries, whose cost follows the region)
    int    xcd_chunk;        // triangle grid: while the pose still moves an XCD's share of a launch is every eighth chunk of this many x 256 queries (0: always one contiguous eighth; oa_kernels.hpp: xcd_block_index_chunked)
};

// host: fill the float fields from h / slack
inline void grid_params_finish(GridParams &gp)
{
    gp.hf = (float)gp.h;
    gp.inv_hf = (float)gp.inv_h;
    const double s = (gp.slack + 4e-6 * gp.h) * (1.0 + 1e-6);
    gp.slackf = (float)s;
    if ((double)gp.slackf < s) gp.slackf = nextafterf(gp.slackf, INFINITY);
}

// ---- the 64-ary box tree (oa_bvh.hpp) as far as the grid search needs it: queries it cannot settle are finished through
// the tree by the wave that owns them, in the same launch ---------------------------------------------------------------
constexpr int BVH_W = 64;
constexpr int BVH_MAX_LEVELS = 6;

struct BvhParams {
    int n_prims;                          // real primitives
    int levels;                           // box levels L >= 1: level 1 = leaf boxes ... level L = top (<= 64 boxes)
    int cnt[BVH_MAX_LEVELS + 1];          // boxes at level l (1..L)
    int off[BVH_MAX_LEVELS + 1];          // offset of level l in the box array, in boxes (each level padded to 64)
    double scale, slack;                  // largest |coordinate| and absolute slack (triangle mode's delta)
};

#if defined(__HIPCC__)

// one wave's scratch for a descent: per level 64 lower bounds, the mask of children still to visit, the node.  Strides in
// elements, so that the grid kernels can lay it over the LDS their (finished) scan no longer needs.
struct BvhLds {
    float *lb0; int lb_stride;
    unsigned long long *mask0; int mask_stride;
    int *node0; int node_stride;
    __device__ __forceinline__ float *lb(int level) const { return lb0 + (long long)level * lb_stride; }
    __device__ __forceinline__ unsigned long long *mask(int level) const { return mask0 + (long long)level * mask_stride; }
    __device__ __forceinline__ int *node(int level) const { return node0 + (long long)level * node_stride; }
};

template <bool TRI>
__device__ __forceinline__ void bvh_wave_query(const BvhParams &bp, const float4 *__restrict__ boxes,
                                               const float4 *__restrict__ prims, const float *p, float cutf, float &best,
                                               uint32_t &bidx, float &bx, float &by, float &bz, const BvhLds &lds, int lane);

__device__ __forceinline__ int grid_cell_coord(double q, double lo, double inv_h, int n)
{
    int k = (int)floor((q - lo) * inv_h);
    k = k < 0 ? 0 : k;
    return k >= n ? n - 1 : k;
}

// distance from coordinate v to the slab [lo + k h, lo + (k+1) h] of cell k (0 inside), shrunk by `slack`
__device__ __forceinline__ double grid_axis_gap(double v, double lo, double h, int k, double slack)
{
    const double a = lo + (double)k * h, b = lo + (double)(k + 1) * h;
    double g = v < a ? a - v : (v > b ? v - b : 0.0);
    g -= slack;
    return g > 0.0 ? g : 0.0;
}

// float square root that never underestimates (hardware sqrt is within 1 ulp; the factor covers it and the cast)
__device__ __forceinline__ float grid_sqrt_up(float x)
{
    // v_sqrt_f32 itself (1 ulp), not the correctly rounded sequence around it (19 instructions per row of cells)
    return __builtin_amdgcn_sqrtf(x) * 1.000001f + 1e-18f;         // + 1e-18: x below the float normal range (sqrt < 1.1e-19)
}

// cell_of: 2 ints per vertex -- its cell and its RANK among the cell's vertices (what the counting atomic returns): the
// scatter then needs no second round of a million atomics on cursors (50 -> ~20 us of a 1M-vertex upload).  The rank is
// the atomics' arrival order -- as arbitrary as the cursors' was; the order inside a cell is immaterial ((d2, index) min).
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_grid_count(const float *__restrict__ xyz, int nt, GridParams gp, int2 *__restrict__ cell_of,
                             int *__restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const int cx = grid_cell_coord((double)xyz[3ll * i], gp.lo[0], gp.inv_h, gp.n[0]);
    const int cy = grid_cell_coord((double)xyz[3ll * i + 1], gp.lo[1], gp.inv_h, gp.n[1]);
    const int cz = grid_cell_coord((double)xyz[3ll * i + 2], gp.lo[2], gp.inv_h, gp.n[2]);
    const int c = (cz * gp.n[1] + cy) * gp.n[0] + cx;
    cell_of[i] = make_int2(c, atomicAdd(&counts[c], 1));
}

// offsets (exclusive scan of counts, long long from k_scan_counts) -> int cell_start; cursor (the triangle grid's fill pass
// counts its cells' entries again): zeroed, or nullptr
__global__ void k_grid_starts(const long long *__restrict__ offsets, int n_cells, int *__restrict__ cell_start,
                              int *__restrict__ cursor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_cells) return;
    cell_start[i] = (int)offsets[i];
    if (cursor && i < n_cells) cursor[i] = 0;
}

__global__ void k_grid_scatter(const float *__restrict__ xyz, int nt, const int2 *__restrict__ cell_of,
                               const int *__restrict__ cell_start, float4 *__restrict__ sorted)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    const int2 cr = cell_of[i];
    const int pos = cell_start[cr.x] + cr.y;
    sorted[pos] = make_float4(xyz[3ll * i], xyz[3ll * i + 1], xyz[3ll * i + 2], __int_as_float(i));
}
#endif  // !OA_FAMILY_TU

// ---- the query's frame in the grid -----------------------------------------------------------------------------------
// Located once per query in double (projection onto the grid's box, own cell, position inside that cell); everything
// per row / per ring afterwards is float arithmetic on three numbers per axis.  All distances derived from it are
// LOWER bounds: the distance from the projected query pc to the slab of cells c + d along an axis is
//     d > 0:  d h - f        d < 0:  (-d - 1) h + f        (f = pc - (lo + c h), the position inside the own cell)
// minus `slackf`, which covers the double-level noise of the build's cell assignment and every float rounding here.
struct GridQuery {
    int   c[3];        // own cell (of the projection onto the box)
    float f[3];        // position of the projection inside the own cell, along each axis
    float off2;        // squared distance query -> box (0 inside)
    bool  finite;
};

__device__ __forceinline__ GridQuery grid_locate(const GridParams &gp, float px, float py, float pz, double *pabs = nullptr)
{
    GridQuery q;
    const double p[3] = { (double)px, (double)py, (double)pz };
    double off2 = 0.0, pa = 0.0;
    q.finite = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double ab = fabs(p[a]);
        if (!(ab < INFINITY)) q.finite = false;
        pa += ab;
        const double pc = fmin(fmax(p[a], gp.lo[a]), gp.hi[a]);    // (a NaN coordinate ends up at lo: the query is not searched anyway)
        const double d = p[a] - pc;
        off2 = fma(d, d, off2);
        // cell = floor((pc - lo) / h), as the build's grid_cell_coord has it (t >= 0: the conversion truncates = floors); the
        // position inside the cell from the same quotient: (t - c) h is pc - (lo + c h) up to 1e-13 h (inv_h h = 1 +- 2^-52
        // on at most 1024 cells), far below what slackf covers
        const double t = (pc - gp.lo[a]) * gp.inv_h;
        const int c = min((int)t, gp.n[a] - 1);
        q.c[a] = c;
        q.f[a] = (float)((t - (double)c) * gp.h);
    }
    q.off2 = (float)off2;
    if (pabs) *pabs = pa;
    return q;
}

// lower bound of the distance from the projected query to the slab of cells c + d along one axis
__device__ __forceinline__ float grid_gap(float f, float h, float slack, int d)
{
    const float fd = (float)d;
    const float g = (d > 0 ? __builtin_fmaf(fd, h, -f) : __builtin_fmaf(-fd - 1.0f, h, f)) - slack;
    return (d != 0 && g > 0.f) ? g : 0.f;
}

// A row of cells (fixed y, z) at squared distance >= row2 from the query, searched for everything within squared
// reach `w2 = reach^2 - row2` along x: how many cells left / right of the own column can still matter (0..r each).
// Over-inclusion is harmless.  r == 1 needs no square root.
__device__ __forceinline__ void grid_row_span(float fx, float h, float inv_h, float slack, float w2, int r, int &dl, int &dr)
{
    if (!(w2 < 3.0e38f)) { dl = r; dr = r; return; }
    if (w2 < 0.f) w2 = 0.f;
    if (r == 1) {
        const float gl = fmaxf(fx - slack, 0.f), gr = fmaxf((h - fx) - slack, 0.f);
        dl = (gl * gl <= w2) ? 1 : 0;
        dr = (gr * gr <= w2) ? 1 : 0;
        return;
    }
    // gap(-d) = (d - 1) h + fx - slack <= sw  <=>  d <= (sw - fx + slack) / h + 1   (and the mirror image to the right)
    // (fx is laundered: the compiler otherwise hoists h - fx out of the ring loop and, at 80 registers, spills it -- one
    //  subtraction in a path only the rings r >= 2 take)
    asm volatile("" : "+v"(fx));
    const float sw = grid_sqrt_up(w2) + slack;
    const float fr = (float)r;
    const float tl = fminf(fmaxf(__builtin_floorf((sw - fx) * inv_h * 1.000001f) + 1.0f, 0.f), fr);
    const float tr = fminf(fmaxf(__builtin_floorf((sw - (h - fx)) * inv_h * 1.000001f) + 1.0f, 0.f), fr);
    dl = (int)tl; dr = (int)tr;
}

// lower bound of the squared distance from the query to anything outside the cube of Chebyshev radius r around the
// own cell (only faces interior to the grid count); +inf when the cube covers the grid
__device__ __forceinline__ float grid_cube_bound2(const GridParams &gp, const GridQuery &q, int r)
{
    float m = INFINITY;
    const float fr = (float)r;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (q.c[a] - r > 0) m = fminf(m, __builtin_fmaf(fr, gp.hf, q.f[a]));                  // f + r h
        if (q.c[a] + r + 1 < gp.n[a]) m = fminf(m, __builtin_fmaf(fr + 1.0f, gp.hf, -q.f[a]));   // (r + 1) h - f
    }
    if (!(m < INFINITY)) return INFINITY;
    m = fmaxf(m - gp.slackf, 0.f);
    return __builtin_fmaf(m, m, q.off2);
}

// Loads through a 32-bit byte offset from a wave-uniform base (global_load ... v_off, s[base]): one VGPR and one shift per
// address where the sign-extended 64-bit form costs two of each -- eighteen addresses are in flight per batch of rows.
// The host builds a grid only for targets whose images stay below 4 GiB (build_grid: GRID_MAX_TARGETS, GRID_MAX_CELLS).
constexpr long long GRID_MAX_TARGETS = (1ll << 28) - 1, GRID_MAX_CELLS = (1ll << 30) - 2;
__device__ __forceinline__ int grid_ld_cell(const int *__restrict__ base, int idx)
{
    return *(const int *)((const char *)base + (unsigned)idx * 4u);
}
__device__ __forceinline__ float4 grid_ld_vertex(const float4 *__restrict__ base, int idx)
{
    return *(const float4 *)((const char *)base + (unsigned)idx * 16u);
}

// ---- safe radii (round 4): when the seed alone settles a query --------------------------------------------------------
// For target t let S(t) = the distance to the nearest OTHER target.  A query p with |p - t| < S(t) / 2 has t as its
// nearest target, strictly: any other t' is at |p - t'| >= |t - t'| - |p - t| > S / 2.  The build stores, per target
// (original index), safe2 = (S / 2)^2 (1 - 1e-4) as a float; the margin covers the float metric's rounding on both sides
// (d2_metric is in difference form: relative error < 3e-7 of the true squared distance), so d2_metric(p, t) < safe2
// implies d2_metric(p, t') > d2_metric(p, t) for every other target -- no tie either: brute force would report exactly
// (d2, t).  S is a LOWER bound: the minimum over the 3 x 3 x 3 block of cells around t's cell, capped by the distance
// to everything outside the block (>= h - slack); 0 = "never" for duplicates, crowded blocks (the build does not walk
// more than SAFE_SCAN_MAX candidates per target) and radii below the float normal range (underflow in the metric).
// The search (k_nn_search_grid) keeps {index, safe2} of a slot's winner next to its winner record (`wsafe`, one gather
// from the radii when the winner changes); an entry whose index is not the seed's says nothing -- other kernels write
// winner records and know nothing of this -- and is brought up to date by the search that finds it so.
constexpr int SAFE_SCAN_MAX = 1024;
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_grid_safe_radius(const float4 *__restrict__ sorted, int nt, GridParams gp, const int *__restrict__ cell_start,
                                   float *__restrict__ safe_by_idx)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nt) return;
    const float4 t = sorted[j];
    const int cx = grid_cell_coord((double)t.x, gp.lo[0], gp.inv_h, gp.n[0]);
    const int cy = grid_cell_coord((double)t.y, gp.lo[1], gp.inv_h, gp.n[1]);
    const int cz = grid_cell_coord((double)t.z, gp.lo[2], gp.inv_h, gp.n[2]);
    const int xa = max(cx - 1, 0), xb = min(cx + 1, gp.n[0] - 1);
    // the nine rows' ranges of `sorted` (cells are x-fastest: three cells of a row are one range), all loads in flight together
    int ka[9], kb[9], total = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int z = cz + r / 3 - 1, y = cy + r % 3 - 1;
        const bool in = z >= 0 && z < gp.n[2] && y >= 0 && y < gp.n[1];
        const int row = in ? (z * gp.n[1] + y) * gp.n[0] : 0;
        ka[r] = in ? grid_ld_cell(cell_start, row + xa) : 0;
        kb[r] = in ? grid_ld_cell(cell_start, row + xb + 1) : 0;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) total += kb[r] - ka[r];
    float out = 0.f;
    if (total <= SAFE_SCAN_MAX) {
        const float cap = fmaxf(gp.hf * 0.999999f - gp.slackf, 0.f);
        float m = cap * cap;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            for (int k = ka[r]; k < kb[r]; ++k) {
                const float4 o = grid_ld_vertex(sorted, k);
                const float d = d2_metric(t.x, t.y, t.z, o.x, o.y, o.z);
                if (d < m && k != j) m = d;                         // (a non-finite neighbour can never win a search: ignored)
            }
        }
        out = 0.25f * m * 0.9999f;
        if (!(out >= 1e-30f)) out = 0.f;
    }
    safe_by_idx[(uint32_t)__float_as_int(t.w)] = out;           // by ORIGINAL index: the searches look it up for a slot's winner
}
#endif  // !OA_FAMILY_TU

__device__ __forceinline__ uint32_t grid_ld_safe(const float *__restrict__ base, int idx)
{
    return *(const uint32_t *)((const char *)base + (unsigned)idx * 4u);
}
__device__ __forceinline__ void grid_st_safe(uint2 *__restrict__ base, int slot, uint32_t idx, uint32_t safe_bits)
{
    *(uint2 *)((char *)base + (unsigned)slot * 8u) = make_uint2(idx, safe_bits);
}

// ---- the 3 x 3 rows around the own cell (ring 1) -- where a seeded query starts and, once the pose has settled, ends ---------
// ONE batch for every L (9 rows, at most 10 ranges), so the row offsets are compile-time constants for L = 1, and everything
// that depends on one axis only is computed once: four gaps, the two squared x-gaps of the row spans, the row strides.  The
// bounds are those of the general row code (the same expressions in the same order) at a third of its instructions -- and
// that code was half of k_nn_search_grid's VALU work at 1M points.  Two passes, so that no more than UR rows' ranges sit in
// registers: holding nine rows' ranges cost the kernel its sixth wave per SIMD, or spills.
struct GridBlock3 {
    unsigned rows;     // bits 0..8: rows (dz + 1) * 3 + (dy + 1) that can matter; 9 / 10: the left / right neighbour of the own
                       // cell, for the own row of a SECOND ring (the own cell was ring 0)
    unsigned ends;     // bit k: row k needs the cell left of the own column, bit 9 + k: the cell right of it
    int row0, sy, sz;  // index of the own row's first cell; cells per step in y, in z
};

// pass 1, no loads.  A row at squared distance row2 is skipped when row2 * kf - prune_abs > prune_lim; its span along x is
// what lies within w2 = reach - row2 * kf (reach = +inf: the whole row).  `first`: ring 1 is where this search started.
template <int L>
__device__ __forceinline__ GridBlock3 grid_block3_select(const GridParams &gp, const GridQuery &q, int sub, bool first,
                                                         float prune_lim, float prune_abs, float reach, float kf)
{
    constexpr int RPL = (9 + L - 1) / L;
    const float h = gp.hf, slack = gp.slackf;
    const float gzm = grid_gap(q.f[2], h, slack, -1), gzp = grid_gap(q.f[2], h, slack, 1);
    const float gym = grid_gap(q.f[1], h, slack, -1), gyp = grid_gap(q.f[1], h, slack, 1);
    const float z2m = __builtin_fmaf(gzm, gzm, q.off2), z2p = __builtin_fmaf(gzp, gzp, q.off2);
    const float gl = fmaxf(q.f[0] - slack, 0.f), gr = fmaxf((h - q.f[0]) - slack, 0.f);     // grid_row_span, r == 1
    const float gl2 = gl * gl, gr2 = gr * gr;
    GridBlock3 b;
    b.rows = 0u; b.ends = 0u;
    b.sy = gp.n[0]; b.sz = gp.n[1] * gp.n[0];
    b.row0 = (q.c[2] * gp.n[1] + q.c[1]) * gp.n[0];
#pragma unroll
    for (int m = 0; m < RPL; ++m) {
        const int kk = sub + L * m;
        if (kk >= 9) continue;
        const int qz = (kk >= 3) + (kk >= 6);
        const int dzi = qz - 1, dyi = kk - 3 * qz - 1;
        const int z = q.c[2] + dzi, y = q.c[1] + dyi;
        if (z < 0 || z >= gp.n[2] || y < 0 || y >= gp.n[1]) continue;
        // every primitive of this row of cells is at real distance^2 >= off2 + gy^2 + gz^2 from the query
        const float zz = dzi < 0 ? z2m : (dzi > 0 ? z2p : q.off2);
        const float gy = dyi < 0 ? gym : (dyi > 0 ? gyp : 0.f);
        const float row2 = __builtin_fmaf(gy, gy, zz);
        if (row2 * kf - prune_abs > prune_lim) continue;
        float w2 = reach - row2 * kf;
        bool dl = true, dr = true;
        if (w2 < 3.0e38f) { w2 = fmaxf(w2, 0.f); dl = gl2 <= w2; dr = gr2 <= w2; }
        if (first || kk != 4) {
            b.rows |= 1u << kk;
            b.ends |= ((unsigned)dl << kk) | ((unsigned)dr << (9 + kk));
        } else {
            if (dl && q.c[0] > 0) b.rows |= 1u << 9;
            if (dr && q.c[0] + 1 < gp.n[0]) b.rows |= 1u << 10;
        }
    }
    return b;
}

// pass 2: takes the next row off b.rows and loads its range [ja, jb) of the cell lists (both 0 when none is left)
__device__ __forceinline__ void grid_block3_next(const GridParams &gp, const GridQuery &q, const int *__restrict__ cell_start,
                                                 GridBlock3 &b, int &ja, int &jb)
{
    ja = jb = 0;
    if (!b.rows) return;
    const int k = __ffs((int)b.rows) - 1;
    b.rows &= b.rows - 1u;
    int row = b.row0, xa, xb;
    if (k < 9) {
        const int qz = (k >= 3) + (k >= 6);
        row += (qz - 1) * b.sz + (k - 3 * qz - 1) * b.sy;
        xa = max(q.c[0] - (int)((b.ends >> k) & 1u), 0);
        xb = min(q.c[0] + (int)((b.ends >> (9 + k)) & 1u), gp.n[0] - 1);
    } else { xa = xb = q.c[0] + (k == 9 ? -1 : 1); }
    ja = grid_ld_cell(cell_start, row + xa); jb = grid_ld_cell(cell_start, row + xb + 1);
}
constexpr int grid_block3_unroll(int L) { return L == 1 ? 4 : (L == 2 ? 3 : 2); }      // rows in flight per lane

// `bj` follows the winner's position in `sorted` (-1: still the seed)
__device__ __forceinline__ void grid_candidate(float px, float py, float pz, const float4 q, int j, float &best,
                                               uint32_t &bidx, int &bj)
{
    const float d = d2_metric(px, py, pz, q.x, q.y, q.z);
    const uint32_t qi = (uint32_t)__float_as_int(q.w);
    if (d < best || (d == best && qi < bidx && d < INFINITY)) { best = d; bidx = qi; bj = j; }
}

// L = 1, 2 or 4 lanes per query.  A thread's time is a chain of dependent memory round trips, and a search of fewer
// than ~300k queries leaves most of the chip's wave slots empty: with L lanes the rows of a ring are dealt out to the
// lanes (row k of a batch goes to lane k mod L), every lane scans its rows with its own running best, and the lanes
// merge (d2, index) lexicographically after every batch -- the same candidates as one lane would see or more (a lane
// prunes with its own, looser, `lim`), so the same answer, in a chain 1/L as long.
//
// Structure (as k_tri_search_grid): ONE wave-uniform loop.  Every trip, each lane that still has work lists the cell
// ranges of its next batch of rows (its own ring, its own batch) on a small per-thread list in LDS, then all lanes walk
// their lists in one flattened loop, four vertices per trip.  No range sits in registers across the scan (the kernel
// fits 6 waves per SIMD), and a wave pays max-over-lanes(vertices) trips instead of the sum over the 18 range slots of
// the longest range in each.
constexpr int GRID_SEGS = 10;      // ranges of one batch: 9 rows of the first block (one each), or as many later rows (two each) as fit

//
// ACC (the loop's iterations): ONE launch does what took three.  (i) A query the rings did not settle -- far from the
// target, or in a crowded cell -- is finished through the box tree right here, by the wave that owns it (one wave per
// query, one query after the other: bvh_wave_query; its scratch lies over the range lists, which are dead by then).
// Round 2 appended such queries to a list with an atomic and launched k_bvh_search behind every grid search, on a list
// that is empty most of the time.  (ii) The pair test and the iteration's fp64 sums are taken in the epilogue: the lane
// holds the query, the winner's coordinates and nothing else is needed -- no second pass over src4 / keys / win
// (k_pair_accumulate, 16 us at 1M points and its own launch).  One row of `partials` per workgroup, fixed order.
// keys[] is not written then: nothing reads it inside the loop.
// BT = threads per workgroup: 256, or 512 for the accumulating variant on large shards -- half as many rows of partials for
// the reduction behind it (docs/HISTORY.md 4.3: 1M points 63.6 -> 60.3 us per iteration), the same six waves per SIMD (three
// workgroups of eight waves per CU instead of six of four); small shards lose with the coarser workgroups (100k: +1.5 us).
// STATS (instrumented build of the accumulating variant, OA_GRID_STATS=1): shader-clock stamps at the phase boundaries and
// candidate counts, one row of GRID_STAT_N counters per wave in `stats`
enum { GRID_STAT_WAVES, GRID_STAT_CYC_TOTAL, GRID_STAT_CYC_PROLOGUE, GRID_STAT_CYC_LIST, GRID_STAT_CYC_SCAN, GRID_STAT_CYC_BOOK,
       GRID_STAT_CYC_FINISH, GRID_STAT_CYC_EPILOGUE, GRID_STAT_LOOP_TRIPS, GRID_STAT_SCAN_TRIPS, GRID_STAT_CANDIDATES,
       GRID_STAT_MAX_LANE_CANDIDATES, GRID_STAT_ACCEPTED, GRID_STAT_CYC_EPI_PAIR, GRID_STAT_CYC_EPI_REDUCE, GRID_STAT_CYC_EPI_BARRIER, GRID_STAT_N };
template <int L, bool ACC = false, int BT = 256, bool STATS = false>
#ifndef OA_GRID_MIN_WAVES
#define OA_GRID_MIN_WAVES 6
#endif
__global__ __launch_bounds__(BT, OA_GRID_MIN_WAVES) void k_nn_search_grid(const DevState *__restrict__ st,
                                                        const float4 *__restrict__ src4, int ns, GridParams gp,
                                                        const int *__restrict__ cell_start,
                                                        const float4 *__restrict__ sorted,
                                                        float4 *__restrict__ win,
                                                        unsigned long long *__restrict__ keys,
                                                        int *__restrict__ todo_list, int *__restrict__ todo_count, int turn,
                                                        BvhParams bp = BvhParams{}, const float4 *__restrict__ boxes = nullptr,
                                                        const float4 *__restrict__ prims = nullptr, NormalTest nrm = NormalTest{},
                                                        double *__restrict__ partials = nullptr,
                                                        unsigned long long *__restrict__ stats = nullptr,
                                                        const float *__restrict__ safe_by_idx = nullptr,
                                                        uint2 *__restrict__ wsafe = nullptr)
{
    long long cyc_t0 = 0, cyc_mark = 0, cyc_prologue = 0, cyc_list = 0, cyc_scan = 0, cyc_book = 0, cyc_finish = 0;
    int n_loop_trips = 0, n_scan_trips = 0, n_cand = 0, n_accepted = 0;
    if (STATS) cyc_t0 = cyc_mark = (long long)__builtin_readcyclecounter();
#define OA_GRID_STAMP(acc) do { if (STATS) { const long long now_ = (long long)__builtin_readcyclecounter(); acc += now_ - cyc_mark; cyc_mark = now_; } } while (0)
    constexpr int RPL = (9 + L - 1) / L;                            // rows per lane and batch
    static_assert(RPL <= GRID_SEGS && (L == 1 || 2 * RPL <= GRID_SEGS), "a batch of rows must fit the per-thread range list");
    if (st->halt) return;
    if (turn >= 0 && (st->tree_turn != 0) != (turn != 0)) return;  // not this kernel's turn (DevState::tree_turn)
    __shared__ int2 seg[GRID_SEGS][BT];
    const int vb = xcd_block_index();                               // workgroup order: one contiguous part of the queries per XCD
    const int gt = vb * (int)blockDim.x + threadIdx.x;
    int i = gt / L;
    const int sub = gt % L;                                         // the L lanes of a query are neighbours in a wave
    const bool alive = i < ns;                                      // (lanes past the last query repeat it, silently)
    if (!alive) i = ns - 1;
    const float4 p4 = src4[i];
    const float4 sw = win[i];                                       // seed: this slot's winner record, see below
    float px, py, pz;
    co_find(st, p4.x, p4.y, p4.z, px, py, pz);                // co_find (general.py:287)

    // seed: this slot's winner record (coordinates + index) of the previous search -- one coalesced load where an
    // index would cost a dependent gather from the caller-ordered target array
    float best = INFINITY;
    uint32_t bidx = IDX_NONE;
    int bj = -1;
    // ... and, when the slot's {index, safe2} entry speaks of this very seed and the query lies inside the seed's safe
    // radius, the seed IS the answer (k_grid_safe_radius): no cell is listed, no vertex scanned.  The lanes of a wave that
    // still search share it with fewer neighbours: fewer lines per load, fewer trips for the longest lane.
    bool accepted = false;
    if (__float_as_int(sw.w) >= 0) {
        const float d = d2_metric(px, py, pz, sw.x, sw.y, sw.z);
        if (d < INFINITY) { best = d; bidx = (uint32_t)__float_as_int(sw.w); }
        if (wsafe) {
            const uint2 ws = *(const uint2 *)((const char *)wsafe + (unsigned)i * 8u);    // (32-bit offset from a uniform base, as grid_ld_vertex)
            accepted = ws.x == (uint32_t)__float_as_int(sw.w) && d < __uint_as_float(ws.y);
        }
    }

    // (a wave whose seeds settled all its queries -- a quarter of them at 1M <-> 1M once the pose has converged -- skips the
    //  query frames and the search radius: fp64 work nothing would read)
    GridQuery q;
    q.c[0] = q.c[1] = q.c[2] = 0; q.f[0] = q.f[1] = q.f[2] = 0.f; q.off2 = 0.f; q.finite = false;
    float cutf = INFINITY;
    if (__any(alive && !accepted)) {
        q = grid_locate(gp, px, py, pz);
        cutf = search_cutoff2(st, px, py, pz);
    }
    const float h = gp.hf, inv_h = gp.inv_hf, slack = gp.slackf;
    // `lim` = what an unseen vertex has to beat: the best so far, or the search radius beyond which the pair test
    // (dist < thresh) fails anyway, whichever is smaller
    float lim = fminf(best, cutf);
    bool settled = accepted, over = false;
    // candidates this query may look at before the tree takes it over (split between its lanes); doubled while the
    // pose still moves by a good part of a cell per iteration (stale seeds: most queries need the second ring) -- see
    // k_tri_search_grid
    int budget = gp.budget;
    int budget_extra = 0;                                            // what the doubling added (wave-uniform)
    {
        const int last = (st->n + 4) % 5;
        const double moved = st->use_target && st->n > 0 ? (st->ring_t[last] + st->ring_r[last] * gp.scale) * st->local_per_world : 0.0;
        if (st->n == 0 || moved > gp.moving_h) budget = gp.budget_moving;
        if (L > 1) budget = budget / L + 8;
        budget_extra = budget - (L > 1 ? gp.budget / L + 8 : gp.budget);
    }
    // With a seed the search starts with the whole 3 x 3 x 3 block as its first "ring": the rows and cells the seed's
    // distance rules out are pruned exactly as they would be one ring later, a query far from its cell's faces still
    // loads its own cell only -- and a query near a face (a quarter of them at 1M <-> 1M) saves the separate pass,
    // i.e. two round trips of the wave it shares with 63 others.
    const int r_start = (bidx != IDX_NONE && gp.seeded_start) ? 1 : 0;
    int r = r_start, b0 = 0, n_seg = 0;
    bool busy = q.finite && alive && !accepted;
    if (STATS) n_accepted = __popcll(__ballot(accepted && alive && sub == 0));
    OA_GRID_STAMP(cyc_prologue);
    while (__any(busy)) {
        bool ring_done = false;
        if (STATS) ++n_loop_trips;
        if (busy && r == 1) {
            // the 3 x 3 block (GridBlock3): which rows can matter, then their ranges, UR rows in flight at a time -- one or two
            // rows survive once the pose has settled
            GridBlock3 blk = grid_block3_select<L>(gp, q, sub, r_start == 1, lim, 1e-30f, lim * 1.00001f + 1e-30f, 0.99999f);
            constexpr int UR = grid_block3_unroll(L);
            while (blk.rows) {
                int ja[UR], jb[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) grid_block3_next(gp, q, cell_start, blk, ja[u], jb[u]);
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    if (jb[u] > ja[u] && budget >= 0) {
                        budget -= jb[u] - ja[u];                         // crowded cells: one wave of the tree search is faster
                        if (budget >= 0) { seg[n_seg][threadIdx.x] = make_int2(ja[u], jb[u]); ++n_seg; }
                    }
                }
            }
            ring_done = true;
        } else if (busy) {
            // The (2r+1)^2 rows (y, z) of the ring (r = 0: the own cell; r >= 2), GR per lane at a time: first the cell ranges of the rows are
            // fetched (independent loads, all in flight together), then their vertices are scanned four per trip.
            constexpr int GR = L == 1 ? 5 : RPL;                    // rows per lane and batch here: five rows' ranges in registers, not nine
            const bool first = (r == r_start);
            const int side = 2 * r + 1, n_rows = side * side;
            const unsigned div_mul = 65536u / (unsigned)side + 1u;  // k / side == (k * div_mul) >> 16 for k < 256, side <= 15
            int ja[GR], jb[GR], jc[GR], jd[GR];                 // row m: vertices [ja, jb) and [jc, jd) of `sorted`
#pragma unroll
            for (int m = 0; m < GR; ++m) {
                ja[m] = jb[m] = jc[m] = jd[m] = 0;
                const int kk = b0 + sub + L * m;
                if (kk >= n_rows) continue;
                const int qz = (int)(((unsigned)kk * div_mul) >> 16);
                const int dzi = qz - r, dyi = kk - qz * side - r;
                const int z = q.c[2] + dzi, y = q.c[1] + dyi;
                if (z < 0 || z >= gp.n[2] || y < 0 || y >= gp.n[1]) continue;
                // every vertex of this row of cells is at real distance^2 >= off2 + gy^2 + gz^2 from the query
                const float gz = grid_gap(q.f[2], h, slack, dzi), gy = grid_gap(q.f[1], h, slack, dyi);
                const float row2 = __builtin_fmaf(gy, gy, __builtin_fmaf(gz, gz, q.off2));
                if (row2 * 0.99999f - 1e-30f > lim) continue;                   // cannot beat or tie
                // cells of the row that can still matter: x-gap^2 <= lim' - row2
                int dl, dr;
                grid_row_span(q.f[0], h, inv_h, slack, lim * 1.00001f + 1e-30f - row2 * 0.99999f, r, dl, dr);
                const int xa = max(q.c[0] - dl, 0), xb = min(q.c[0] + dr, gp.n[0] - 1);
                const int row = (z * gp.n[1] + y) * gp.n[0];
                // interior rows were fully covered by ring r-1: only their two end cells are new
                const bool shell_row = first || dzi == -r || dzi == r || dyi == -r || dyi == r;
                if (shell_row) {
                    ja[m] = grid_ld_cell(cell_start, row + xa); jb[m] = grid_ld_cell(cell_start, row + xb + 1);
                } else {
                    const int xl = q.c[0] - r, xr = q.c[0] + r;
                    if (dl == r && xl >= 0) { ja[m] = grid_ld_cell(cell_start, row + xl); jb[m] = grid_ld_cell(cell_start, row + xl + 1); }
                    if (dr == r && xr < gp.n[0]) { jc[m] = grid_ld_cell(cell_start, row + xr); jd[m] = grid_ld_cell(cell_start, row + xr + 1); }
                }
            }
            int consumed = GR;                                    // rows of this batch that went on the list (L == 1: as many as fit)
            bool full = false;
#pragma unroll
            for (int m = 0; m < GR; ++m) {
                if (L == 1 && !full && n_seg + 2 > GRID_SEGS) { full = true; consumed = m; }
                if (full) continue;
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    const int j0 = sg ? jc[m] : ja[m], j1 = sg ? jd[m] : jb[m];
                    if (j1 > j0 && budget >= 0) {
                        budget -= j1 - j0;                           // crowded cells: one wave of the tree search is faster
                        if (budget >= 0) { seg[n_seg][threadIdx.x] = make_int2(j0, j1); ++n_seg; }
                    }
                }
            }
            b0 += consumed * L;
            ring_done = b0 >= n_rows;
        }
        OA_GRID_STAMP(cyc_list);
        {   // every lane walks ITS ranges, four vertices per trip (the clamped repeats of the last vertex change nothing)
            int k = 0, j = 0, end = 0;
            while (true) {
                if (j >= end && k < n_seg) { const int2 sgm = seg[k][threadIdx.x]; j = sgm.x; end = sgm.y; ++k; }
                const bool active = j < end;
                if (!__any(active)) break;
                if (STATS) { ++n_scan_trips; if (active) n_cand += min(4, end - j); }
                if (active) {
                    const int last = end - 1;
                    const int e1 = min(j + 1, last), e2 = min(j + 2, last), e3 = min(j + 3, last);
                    const float4 q0 = grid_ld_vertex(sorted, j), q1 = grid_ld_vertex(sorted, e1), q2 = grid_ld_vertex(sorted, e2), q3 = grid_ld_vertex(sorted, e3);
                    grid_candidate(px, py, pz, q0, j, best, bidx, bj);
                    grid_candidate(px, py, pz, q1, e1, best, bidx, bj);
                    grid_candidate(px, py, pz, q2, e2, best, bidx, bj);
                    grid_candidate(px, py, pz, q3, e3, best, bidx, bj);
                    j += 4;
                }
            }
            n_seg = 0;
        }
        OA_GRID_STAMP(cyc_scan);
        over = busy && budget < 0;
        if (L > 1) {                                             // the lanes of the query agree on the best so far
#pragma unroll
            for (int o = 1; o < L; o <<= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const uint32_t oi = (uint32_t)__shfl_xor((int)bidx, o, 64);
                const int oj = __shfl_xor(bj, o, 64);
                if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; bj = oj; }
                over = (__shfl_xor((int)over, o, 64) != 0) || over;
            }
        }
        lim = fminf(best, cutf);
        if (busy) {
            if (over) busy = false;
            else if (ring_done) {
                // lower bound for everything outside the cube of radius r (+inf: the cube covers the whole grid)
                const float bound = grid_cube_bound2(gp, q, r);
                if (!(bound < INFINITY) || bound * 0.99999f - 1e-30f > lim) { settled = true; busy = false; }   // no unseen vertex can beat or tie, or matter
                else { ++r; b0 = 0; if (r > gp.r_max) busy = false; }
            }
        }
        OA_GRID_STAMP(cyc_book);
    }
    bool tight = budget < budget_extra;                             // used more than the base budget allows
    if (L > 1) {
#pragma unroll
        for (int o = 1; o < L; o <<= 1) tight = (__shfl_xor((int)tight, o, 64) != 0) || tight;
    }
    if (!ACC) {
        if (sub != 0 || !alive) return;
        keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
        // the winner record is read by k_pair_accumulate, the tree search and the next search; a winner that is still the
        // seed (the usual case once the loop converges) is already there
        if (bj >= 0) win[i] = sorted[bj];
        // the winner's safe radius beside its record: a new winner's, or the seed's when the slot's entry spoke of another vertex
        // (the entry's index is read again here rather than a flag held through the scan)
        if (wsafe && bidx != IDX_NONE && grid_ld_safe((const float *)wsafe, 2 * i) != bidx) grid_st_safe(wsafe, i, bidx, grid_ld_safe(safe_by_idx, (int)bidx));
        if (!settled) todo_list[atomicAdd(todo_count, 1)] = i;      // finished exactly by the tree search (k_bvh_search)
        // How crowded the hand-over is per wave: what the host looks at before it lets a later search finish its own leftovers
        // (grid_fast_now).  Counted against the BASE budget: a search that ran on the doubled one -- the first of a loop, or
        // while the pose moves -- would otherwise report "nothing handed over" for queries the next search, on the base
        // budget, hands over by the thousand (seen on C5's surface clouds: one 0.6 ms iteration per loop).
        if (!settled || tight) atomicMax(todo_count + 1, __popcll(__ballot(1)));
        return;
    }
    // ---- ACC: finish, record, accumulate -- all threads stay to the end (wave-wide descents, workgroup-wide reduction)
    __shared__ double red[BT / 64][NSUMS];
    const bool mine = sub == 0 && alive;
    // the source point again (a coalesced, cached 16-byte load) rather than three registers held through the whole scan:
    // the pointer is laundered so that the compiler does not merge this load with the one at the top.  Issued here, with
    // the winner's record, so that the two round trips overlap (and the pair test hides what is left of this one)
    const float4 *src_again = src4;
    asm volatile("" : "+s"(src_again));
    const float4 a4 = src_again[i];
    // ... and the index of the slot's safe-radius entry (rather than a flag held through the scan)
    uint32_t ws_idx = IDX_NONE;
    if (wsafe) ws_idx = grid_ld_safe((const float *)wsafe, 2 * i);
    float4 wq = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));  // the winner's record: the seed's, or the scan's
    if (mine && bidx != IDX_NONE) wq = bj >= 0 ? sorted[bj] : win[i];      // (bj < 0: still the seed, re-read rather than kept in registers through the scan)
    bool changed = mine && bj >= 0;
    {
        // (a non-finite query has no finite distance: it stays as it is; recomputed here, not held through the scan)
        const bool finite = fabsf(px) < INFINITY && fabsf(py) < INFINITY && fabsf(pz) < INFINITY;
        unsigned long long todo = __ballot(mine && !settled && finite);
        const unsigned long long crowd = __ballot(mine && finite && (!settled || tight));   // (against the base budget, see above)
        if (crowd && (threadIdx.x & 63) == 0) atomicMax(todo_count + 1, __popcll(crowd));
        if (todo) {
            if ((threadIdx.x & 63) == 0) atomicAdd(todo_count, __popcll(todo));
            // this wave's columns of the range lists are free now: per level 256 B of bounds, then the mask and the node
            const int lane = threadIdx.x & 63, col0 = threadIdx.x & ~63;
            char *base = (char *)&seg[0][col0];
            const int row_bytes = (int)sizeof(seg[0]);
            const BvhLds lds{ (float *)base, row_bytes / 4, (unsigned long long *)(base + 256), row_bytes / 8, (int *)(base + 264), row_bytes / 4 };
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const float qp[3] = { __shfl(px, l, 64), __shfl(py, l, 64), __shfl(pz, l, 64) };
                float b = __shfl(best, l, 64);
                uint32_t bi = (uint32_t)__shfl((int)bidx, l, 64);
                float tx = __shfl(wq.x, l, 64), ty = __shfl(wq.y, l, 64), tz = __shfl(wq.z, l, 64);
                const uint32_t bi0 = bi;
                bvh_wave_query<false>(bp, boxes, prims, qp, __shfl(cutf, l, 64), b, bi, tx, ty, tz, lds, lane);
                if (lane == l && bi != bi0) { best = b; bidx = bi; wq = make_float4(tx, ty, tz, __int_as_float((int)bi)); changed = true; }
            }
        }
    }
    if (changed) win[i] = wq;                                     // the next search's seed (and what a one-shot call would read)
    // the winner's safe radius beside its record: a new winner's, or the seed's when the slot's entry spoke of another vertex
    if (wsafe && mine && bidx != IDX_NONE && ws_idx != bidx) grid_st_safe(wsafe, i, bidx, grid_ld_safe(safe_by_idx, (int)bidx));
    OA_GRID_STAMP(cyc_finish);
    bool valid = false;
    float vbx = 0.f, vby = 0.f, vbz = 0.f;
    double dist = 0.0;
    if (mine && bidx != IDX_NONE) {
        float tn[3] = { 0.f, 0.f, 0.f };
        if (nrm.src_n) { tn[0] = nrm.tgt_n[3ll * bidx]; tn[1] = nrm.tgt_n[3ll * bidx + 1]; tn[2] = nrm.tgt_n[3ll * bidx + 2]; }
        valid = pair_eval(st, px, py, pz, wq.x, wq.y, wq.z, nrm, i, tn, st->thresh, vbx, vby, vbz, dist);
    }
    const double pvx = st->pivot[0], pvy = st->pivot[1], pvz = st->pivot[2];
    long long epi_stamps[2] = { 0, 0 };
    const long long cyc_pair = STATS ? (long long)__builtin_readcyclecounter() : 0;     // loads + pair test done (epilogue started at cyc_mark)
    block_store_pair(valid, (double)a4.x - pvx, (double)a4.y - pvy, (double)a4.z - pvz, (double)vbx - pvx, (double)vby - pvy,
                     (double)vbz - pvz, dist - st->d_pivot, red, partials + (long long)vb * NSUMS, STATS ? epi_stamps : nullptr);
    if (STATS && stats) {
        const long long now = (long long)__builtin_readcyclecounter();
        int sum = n_cand, mx = n_cand;
        for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); mx = max(mx, __shfl_xor(mx, o, 64)); }
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *row = stats + ((size_t)blockIdx.x * (BT / 64) + (threadIdx.x >> 6)) * GRID_STAT_N;
            row[GRID_STAT_WAVES] = 1ull;
            row[GRID_STAT_CYC_TOTAL] = (unsigned long long)(now - cyc_t0);
            row[GRID_STAT_CYC_PROLOGUE] = (unsigned long long)cyc_prologue;
            row[GRID_STAT_CYC_LIST] = (unsigned long long)cyc_list;
            row[GRID_STAT_CYC_SCAN] = (unsigned long long)cyc_scan;
            row[GRID_STAT_CYC_BOOK] = (unsigned long long)cyc_book;
            row[GRID_STAT_CYC_FINISH] = (unsigned long long)cyc_finish;
            row[GRID_STAT_CYC_EPILOGUE] = (unsigned long long)(now - cyc_mark);
            row[GRID_STAT_LOOP_TRIPS] = (unsigned long long)n_loop_trips;
            row[GRID_STAT_SCAN_TRIPS] = (unsigned long long)n_scan_trips;
            row[GRID_STAT_CANDIDATES] = (unsigned long long)sum;
            row[GRID_STAT_MAX_LANE_CANDIDATES] = (unsigned long long)mx;
            row[GRID_STAT_ACCEPTED] = (unsigned long long)n_accepted;
            row[GRID_STAT_CYC_EPI_PAIR] = (unsigned long long)(cyc_pair - cyc_mark);
            row[GRID_STAT_CYC_EPI_REDUCE] = (unsigned long long)(epi_stamps[0] - cyc_pair);
            row[GRID_STAT_CYC_EPI_BARRIER] = (unsigned long long)(epi_stamps[1] - epi_stamps[0]);
        }
    }
#undef OA_GRID_STAMP
}

// non-zero entries of a[0 .. n): one count per WORKGROUP of a grid-stride launch, out[blockIdx.x] -- the host adds them up
// (they go straight to mapped host memory, oa_icp.hip: result_buffer).  History: one atomic per wave -- 8000 of them on one
// word for the 500k cells of a 1M-vertex target -- was 92 us of a 0.8 ms target upload; one per workgroup 8 us; none now.
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_count_nonzero(const int *__restrict__ a, int n, int *__restrict__ out)
{
    __shared__ int part[16];
    int cnt = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) cnt += a[i] != 0 ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += part[w];
        out[blockIdx.x] = sum;
    }
}
#endif  // !OA_FAMILY_TU

#endif  // __HIPCC__
}  // namespace oa
