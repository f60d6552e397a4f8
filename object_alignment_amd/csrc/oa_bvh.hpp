// oa_bvh.hpp -- exact nearest vertex / nearest triangle through a 64-ary bounding-box tree, one WAVE per query.
//
// Why it exists: the uniform-grid searches (oa_grid.hpp, oa_tri.hpp) settle a query only when its nearest
// primitive lies within `r_max` rings of cells.  Partial overlaps, holes and the first iterations of a badly
// aligned pair leave queries far from the surface; finishing those by brute force costs O(N_target) each.  This tree
// finishes them in O(log) steps and is exact for any input.  (The reference leaves all of this to Blender's BVHTree,
// functions/general.py:297; this is the wave64 restatement of that idea, not of Blender's code.)
//
// Layout: primitives sorted by the 30-bit Morton code of their centre; leaf j = sorted primitives [64 j, 64 j + 64);
// level-1 box j = bounds of leaf j, level-(l+1) box k = bounds of level-l boxes [64 k, 64 k + 64), up to a top level
// of <= 64 boxes.  A wave handles one query: its 64 lanes test the 64 children of a node (or the 64 primitives of a
// leaf) in ONE step with coalesced loads and no divergence; the running best (d2, index) is wave-uniform.  Children
// are visited nearest-first; the per-level candidate masks and bounds live in LDS (6 levels x 64 lanes per wave).
//
// Exactness: candidates are evaluated with the same float32 arithmetic as the brute-force kernels on the ORIGINAL
// coordinates and merged lexicographically on (d2, original index); a box is skipped only if no primitive inside
// can beat OR tie the best: lb (1 - 1e-5) - 1e-30 > best, where lb is a lower bound of the squared distance to the
// box (vertex mode: float32 gaps, relative error < 6u, metric >= D (1 - 5.01u); triangle mode: double, minus
// delta = 64u(|coords|) for the float32 closest-point evaluation, as in oa_tri.hpp).
#pragma once
#include "oa_tri.hpp"

namespace oa {

// (BVH_W, BVH_MAX_LEVELS, BvhParams, BvhLds and the declaration of bvh_wave_query live in oa_grid.hpp: the grid search
//  finishes the queries it cannot settle through this tree, in the same launch)

#if defined(__HIPCC__)

// ---- wave-wide unsigned minimum, result uniform (DPP inside rows of 16, readlane across the 4 rows)
__device__ __forceinline__ uint32_t dpp_min_step(uint32_t v, const int ctrl_sel)
{
    uint32_t o;
    switch (ctrl_sel) {
    case 0:  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
    case 1:  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
    case 2:  o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false); break;  // row_half_mirror
    default: o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false); break;  // row_mirror
    }
    return o < v ? o : v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = dpp_min_step(v, 0);
    v = dpp_min_step(v, 1);
    v = dpp_min_step(v, 2);
    v = dpp_min_step(v, 3);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// ---- build ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned bvh_spread10(unsigned v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Morton key of a primitive's centre (vertex: the point; triangle: centre of its bounding box)
template <bool TRI>
__global__ void k_bvh_keys(const float *__restrict__ xyz, const float4 *__restrict__ tri9, int n, float lx, float ly,
                           float lz, float sx, float sy, float sz, unsigned *__restrict__ keys, int *__restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float cx, cy, cz;
    if (TRI) {
        float a[3], b[3], c[3];
        load_tri(tri9, i, a, b, c);
        cx = 0.5f * (fminf(fminf(a[0], b[0]), c[0]) + fmaxf(fmaxf(a[0], b[0]), c[0]));
        cy = 0.5f * (fminf(fminf(a[1], b[1]), c[1]) + fmaxf(fmaxf(a[1], b[1]), c[1]));
        cz = 0.5f * (fminf(fminf(a[2], b[2]), c[2]) + fmaxf(fmaxf(a[2], b[2]), c[2]));
    } else {
        cx = xyz[3ll * i]; cy = xyz[3ll * i + 1]; cz = xyz[3ll * i + 2];
    }
    const float fx = fminf(fmaxf((cx - lx) * sx, 0.f), 1023.f), fy = fminf(fmaxf((cy - ly) * sy, 0.f), 1023.f),
                fz = fminf(fmaxf((cz - lz) * sz, 0.f), 1023.f);      // NaN -> 0 through fmaxf
    keys[i] = bvh_spread10((unsigned)fx) | (bvh_spread10((unsigned)fy) << 1) | (bvh_spread10((unsigned)fz) << 2);
    ids[i] = i;
}

// sorted primitive images.  vertex: float4 {x, y, z, bits(original index)}; triangle: the tri9 image with the original
// index in the second lane of its third float4.  Slots past n are NaN (never selected) with index IDX_NONE.
template <bool TRI>
__global__ void k_bvh_gather(const float *__restrict__ xyz, const float4 *__restrict__ tri9, const int *__restrict__ order,
                             int n, int n_pad, float4 *__restrict__ prims)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pad) return;
    if (TRI) {
        if (j < n) {
            const int t = order[j];
            float4 w = tri9[3ll * t + 2];
            w.y = __uint_as_float((uint32_t)t);
            prims[3ll * j] = tri9[3ll * t]; prims[3ll * j + 1] = tri9[3ll * t + 1]; prims[3ll * j + 2] = w;
        } else {
            const float4 bad = make_float4(NAN, NAN, NAN, NAN);
            prims[3ll * j] = bad; prims[3ll * j + 1] = bad;
            prims[3ll * j + 2] = make_float4(NAN, __uint_as_float(IDX_NONE), 0.f, 0.f);
        }
    } else {
        if (j < n) {
            const int v = order[j];
            prims[j] = make_float4(xyz[3ll * v], xyz[3ll * v + 1], xyz[3ll * v + 2], __uint_as_float((uint32_t)v));
        } else {
            prims[j] = make_float4(NAN, NAN, NAN, __uint_as_float(IDX_NONE));
        }
    }
}

// min / max of a box's six numbers over the 64 lanes of a wave (fminf / fmaxf ignore NaN and are exact: the result does not
// depend on the order -- what a serial loop over the 64 elements gives, up to the sign of a zero); lane 0 writes the box
__device__ __forceinline__ void bvh_wave_box_store(float lo[3], float hi[3], float4 *__restrict__ boxes, long long b)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        boxes[2 * b] = make_float4(lo[0], lo[1], lo[2], 0.f);
        boxes[2 * b + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

// level-1 boxes: bounds of 64 consecutive primitives (NaN coordinates are ignored by fminf/fmaxf; an all-NaN leaf gets
// the empty box lo = +inf, hi = -inf).  One WAVE per box, lane k takes primitive k -- one thread per box walked its 64
// primitives one dependent load after the other (31 us for 82k triangles, on the critical path of every target upload).
// Launch: 256 threads, (n_boxes + 3) / 4 workgroups.
template <bool TRI>
__global__ void k_bvh_leaf_boxes(const float4 *__restrict__ prims, int n_pad, int n_boxes, float4 *__restrict__ boxes)
{
    const long long b = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_boxes) return;                                       // (wave-uniform)
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    const long long j = b * BVH_W + (threadIdx.x & 63);
    if (j < n_pad) {
        if (TRI) {
            float p[3][3];
            load_tri(prims, j, p[0], p[1], p[2]);
            for (int v = 0; v < 3; ++v)
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[v][a]); hi[a] = fmaxf(hi[a], p[v][a]); }
        } else {
            const float4 q = prims[j];
            lo[0] = fminf(lo[0], q.x); lo[1] = fminf(lo[1], q.y); lo[2] = fminf(lo[2], q.z);
            hi[0] = fmaxf(hi[0], q.x); hi[1] = fmaxf(hi[1], q.y); hi[2] = fmaxf(hi[2], q.z);
        }
    }
    bvh_wave_box_store(lo, hi, boxes, b);
}

// level l boxes from the 64 boxes below each; one wave per box, same launch shape
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_bvh_upper_boxes(const float4 *__restrict__ child, int n_child, int n_boxes, float4 *__restrict__ boxes)
{
    const long long b = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_boxes) return;                                       // (wave-uniform)
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    const long long j = b * BVH_W + (threadIdx.x & 63);
    if (j < n_child) {
        const float4 l = child[2 * j], h = child[2 * j + 1];
        lo[0] = fminf(lo[0], l.x); lo[1] = fminf(lo[1], l.y); lo[2] = fminf(lo[2], l.z);
        hi[0] = fmaxf(hi[0], h.x); hi[1] = fmaxf(hi[1], h.y); hi[2] = fmaxf(hi[2], h.z);
    }
    bvh_wave_box_store(lo, hi, boxes, b);
}
#endif  // !OA_FAMILY_TU

// ---- query ---------------------------------------------------------------------------------------------------------
// lower bound of the squared distance from p to anything inside the box, as a float that never exceeds the real bound
// by more than the (1 - 1e-5) factor of the prune test absorbs
template <bool TRI>
__device__ __forceinline__ float bvh_box_bound(const float *p, const float4 lo, const float4 hi, double delta)
{
    if (TRI) {
        const double gx = fmax(fmax((double)lo.x - (double)p[0], (double)p[0] - (double)hi.x), 0.0);
        const double gy = fmax(fmax((double)lo.y - (double)p[1], (double)p[1] - (double)hi.y), 0.0);
        const double gz = fmax(fmax((double)lo.z - (double)p[2], (double)p[2] - (double)hi.z), 0.0);
        double lb = sqrt(gx * gx + gy * gy + gz * gz) - delta;
        lb = lb > 0.0 ? lb : 0.0;
        return (float)(lb * lb * (1.0 - 1e-6));                  // the float rounding stays below the 1e-6
    } else {
        const float gx = fmaxf(fmaxf(lo.x - p[0], p[0] - hi.x), 0.f);
        const float gy = fmaxf(fmaxf(lo.y - p[1], p[1] - hi.y), 0.f);
        const float gz = fmaxf(fmaxf(lo.z - p[2], p[2] - hi.z), 0.f);
        return gx * gx + gy * gy + gz * gz;
    }
}

__device__ __forceinline__ bool bvh_prune(float lb, float best)
{
    return lb * 0.99999f - 1e-30f > best;                          // cannot beat or tie
}

// The descent for ONE query by ONE wave (all 64 lanes call it together; p, cutf and the running best are wave-uniform).
// In: the seed (best, bidx; vertex mode: its coordinates in bx, by, bz).  Out: the exact nearest primitive -- the same
// (d2, index) every other search returns.  lds: this wave's scratch (per level 64 bounds, a mask, a node).
template <bool TRI>
__device__ __forceinline__ void bvh_wave_query(const BvhParams &bp, const float4 *__restrict__ boxes,
                                               const float4 *__restrict__ prims, const float *p, float cutf, float &best,
                                               uint32_t &bidx, float &bx, float &by, float &bz, const BvhLds &lds, int lane)
{
    const int top = bp.levels;
    // `lim`: the best so far or the search radius (search_cutoff2), whichever is smaller -- see k_nn_search_grid
    float lim = fminf(best, cutf);
    const bool finite = fabsf(p[0]) < INFINITY && fabsf(p[1]) < INFINITY && fabsf(p[2]) < INFINITY;
    if (!finite) return;                                            // a non-finite query has no finite distance: stays as seeded
    const double delta = TRI ? 64.0 * 5.9604644775390625e-08 * (bp.scale + fabs((double)p[0]) + fabs((double)p[1]) + fabs((double)p[2])) + bp.slack : 0.0;
    float thr = TRI ? tri_skip_threshold(lim, delta) : 0.f;         // squared bounding-box gap beyond which a triangle is out
    int level = top;
    int node = 0;
    bool fresh = true;                                              // `level` holds a node whose children are not tested yet
    while (true) {
        if (fresh) {
            const long long ch = (long long)node * BVH_W + lane;
            float lb = INFINITY;
            bool pass = false;
            if (ch < bp.cnt[level]) {
                const float4 lo = boxes[2 * ((long long)bp.off[level] + ch)], hi = boxes[2 * ((long long)bp.off[level] + ch) + 1];
                lb = bvh_box_bound<TRI>(p, lo, hi, delta);
                pass = lb < INFINITY && !bvh_prune(lb, lim);
            }
            lds.lb(level)[lane] = lb;
            const unsigned long long m = __ballot(pass);
            if (lane == 0) { *lds.mask(level) = m; *lds.node(level) = node; }
            fresh = false;
        }
        const unsigned long long m = *lds.mask(level);
        if (m == 0ull) {
            if (level == top) break;
            ++level;
            continue;
        }
        const bool member = (m >> lane) & 1ull;
        const uint32_t lbits = member ? __float_as_uint(lds.lb(level)[lane]) : 0xFFFFFFFFu;   // bounds are >= +0
        const uint32_t mb = wave_min_u32(lbits);
        if (bvh_prune(__uint_as_float(mb), lim)) {                  // the nearest candidate is out: so are the others
            if (lane == 0) *lds.mask(level) = 0ull;
            continue;
        }
        const unsigned long long eq = __ballot(member && lbits == mb);
        const int pick = __ffsll((long long)eq) - 1;
        if (lane == 0) *lds.mask(level) = m & ~(1ull << pick);
        const long long child = (long long)*lds.node(level) * BVH_W + pick;
        if (level > 1) {
            --level;
            node = (int)child;
            fresh = true;
            continue;
        }
        // leaf: 64 primitives, one per lane
        const long long j = child * BVH_W + lane;
        float d;
        uint32_t qi;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TRI) {
            float a[3], b[3], c[3], r[3];
            load_tri(prims, j, a, b, c);
            qi = __float_as_uint(prims[3 * j + 2].y);
            // the leaf's box passed, but it bounds 64 triangles: when no single triangle's own box can matter
            // the ~300-instruction closest-point evaluation of the whole wave is skipped
            float lb = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float lo = fminf(fminf(a[k], b[k]), c[k]), hi = fmaxf(fmaxf(a[k], b[k]), c[k]);
                const float g = fmaxf(fmaxf(lo - p[k], p[k] - hi), 0.f);
                lb += g * g;
            }
            if (!__any(!(lb > thr))) continue;
            closest_on_tri(p, a, b, c, r);
            d = tri_dist2(p, r);
        } else {
            q = prims[j];
            qi = __float_as_uint(q.w);
            d = d2_metric(p[0], p[1], p[2], q.x, q.y, q.z);
        }
        const uint32_t dbits = (d < INFINITY) ? __float_as_uint(d) : 0xFFFFFFFFu;   // d >= +0 here; NaN / inf never win
        const uint32_t md = wave_min_u32(dbits);
        if (md <= __float_as_uint(best) && md != 0xFFFFFFFFu) {
            const uint32_t mi = wave_min_u32(dbits == md ? qi : IDX_NONE);
            if (md < __float_as_uint(best) || mi < bidx) {
                best = __uint_as_float(md); bidx = mi; lim = fminf(best, cutf);
                if (TRI) thr = tri_skip_threshold(lim, delta);
                else {
                    const int wl = __ffsll((long long)__ballot(dbits == md && qi == mi)) - 1;   // the winner's lane
                    bx = __shfl(q.x, wl, 64); by = __shfl(q.y, wl, 64); bz = __shfl(q.z, wl, 64);
                }
            }
        }
    }
}

// One wave per query.  list == nullptr: every source point, seeded with last iteration's primitive (vertex mode: the
// winner record win[i]; triangles: `prev`, original index, evaluated through tri9); else the points the grid search
// could not settle, seeded with keys[i].
// ACC (loop iterations, whole-shard searches only): the pair test and the fp64 sums of the iteration are taken in the
// same launch -- the wave holds its query, the winner and d2 already; every wave keeps running sums over its queries
// (wave-uniform, so no cross-lane reduction), the workgroup's four waves are added in order into one row of `partials`.
// keys[] / prev[] are not written then: nothing reads them inside the loop.
// ACC launches workgroups of 16 waves (one row of partials per 16 queries in flight, not per 4: the single-workgroup
// reduction behind it is a chain of row loads).
template <bool TRI, bool ACC = false>
__global__ __launch_bounds__(ACC ? 1024 : 256) void k_bvh_search(const DevState *__restrict__ st, const float4 *__restrict__ src4,
                                                    int ns, BvhParams bp, const float4 *__restrict__ boxes,
                                                    const float4 *__restrict__ prims,
                                                    const float4 *__restrict__ tri9,
                                                    int *__restrict__ prev, float4 *__restrict__ win,
                                                    unsigned long long *__restrict__ keys,
                                                    const int *__restrict__ list, const int *__restrict__ list_count, int turn,
                                                    NormalTest nrm = NormalTest{}, double *__restrict__ partials = nullptr,
                                                    const float *__restrict__ safe_by_idx = nullptr, uint2 *__restrict__ wsafe = nullptr)
{
    // safe_by_idx / wsafe (whole-shard vertex searches): a seed inside its SAFE RADIUS is the answer, the descent is skipped
    // (k_grid_safe_radius, oa_grid.hpp) -- one wave per query here, so every accepted query saves its wave the whole walk
    if (st->halt) return;
    if (turn >= 0 && (st->tree_turn != 0) != (turn != 0)) return;  // not this kernel's turn (DevState::tree_turn)
    constexpr int WPB = ACC ? 16 : 4;                               // waves per workgroup
    __shared__ float s_lb[WPB][BVH_MAX_LEVELS + 1][BVH_W];
    __shared__ unsigned long long s_mask[WPB][BVH_MAX_LEVELS + 1];
    __shared__ int s_node[WPB][BVH_MAX_LEVELS + 1];
    __shared__ double red[WPB][NSUMS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const BvhLds lds{ &s_lb[w][0][0], BVH_W, &s_mask[w][0], 1, &s_node[w][0], 1 };
    const int n_items = list ? *list_count : ns;
    const int n_waves = gridDim.x * WPB;
    // ACC: the wave's running sums: lane k owns number k, ONE double per lane (all 24 in every lane would sit in registers
    // through the whole descent and halve the kernel's occupancy; in LDS every query paid 20 read-modify-write round trips)
    double lane_sum = 0.0;
    const double thresh = st->thresh, pvx = st->pivot[0], pvy = st->pivot[1], pvz = st->pivot[2], d_pivot = st->d_pivot;

    for (int slot = blockIdx.x * WPB + w; slot < n_items; slot += n_waves) {
        const int i = list ? list[slot] : slot;
        const float4 p4 = src4[i];
        float p[3];
        co_find(st, p4.x, p4.y, p4.z, p[0], p[1], p[2]);          // co_find (general.py:287)

        float best = INFINITY;
        uint32_t bidx = IDX_NONE;
        // vertex mode: (bx, by, bz) follow the winner's coordinates; win[i] is the slot's winner record (coordinates +
        // index) -- the seed of a whole search, the grid search's partial answer in list mode
        float bx = 0.f, by = 0.f, bz = 0.f;
        bool accepted = false;
        uint32_t ws_idx = IDX_NONE;                                 // whose radius the slot's entry holds
        if (list) {
            const unsigned long long k0 = keys[i];
            const float b0 = __uint_as_float((uint32_t)(k0 >> 32));
            if (b0 < INFINITY) {
                best = b0; bidx = (uint32_t)k0;
                if (!TRI) { const float4 sw = win[i]; bx = sw.x; by = sw.y; bz = sw.z; }
            }
        } else if (TRI) {
            const int s = prev ? prev[i] : -1;
            if (s >= 0) {
                float a[3], b[3], c[3], r[3];
                load_tri(tri9, s, a, b, c);
                closest_on_tri(p, a, b, c, r);
                const float d = tri_dist2(p, r);
                if (d < INFINITY) { best = d; bidx = (uint32_t)s; }
            }
        } else {
            const float4 sw = win[i];
            if (__float_as_int(sw.w) >= 0) {
                const float d = d2_metric(p[0], p[1], p[2], sw.x, sw.y, sw.z);
                if (d < INFINITY) { best = d; bidx = (uint32_t)__float_as_int(sw.w); bx = sw.x; by = sw.y; bz = sw.z; }
                if (wsafe) {
                    const uint2 ws = wsafe[i];
                    ws_idx = ws.x;
                    accepted = ws.x == (uint32_t)__float_as_int(sw.w) && d < __uint_as_float(ws.y);
                }
            }
        }
        if (!accepted) {
            const float cutf = search_cutoff2(st, p[0], p[1], p[2]);
            bvh_wave_query<TRI>(bp, boxes, prims, p, cutf, best, bidx, bx, by, bz, lds, lane);
        }
        // the winner's safe radius beside its record: a new winner's, or the seed's when the slot's entry spoke of another vertex
        if (!TRI && !list && wsafe && lane == 0 && bidx != IDX_NONE && ws_idx != bidx) wsafe[i] = make_uint2(bidx, __float_as_uint(safe_by_idx[bidx]));
        if (ACC) {
            if (lane == 0 && !accepted) {
                if (TRI) prev[i] = (bidx == IDX_NONE) ? -1 : (int)bidx;      // the next search's seed
                else win[i] = make_float4(bx, by, bz, __int_as_float((int)bidx));
            }
            if (bidx != IDX_NONE) {                                 // wave-uniform: every lane adds the same terms to its copy
                float qx = bx, qy = by, qz = bz;
                float tn[3] = { 0.f, 0.f, 0.f };
                if (TRI) {
                    float ta[3], tb[3], tc[3], rr[3];
                    load_tri(tri9, bidx, ta, tb, tc);
                    closest_on_tri(p, ta, tb, tc, rr);
                    qx = rr[0]; qy = rr[1]; qz = rr[2];
                    if (nrm.src_n) {                                 // geometric face normal (Blender normal_tri_v3 order)
                        const float e1[3] = { ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2] };
                        const float e2[3] = { tb[0] - tc[0], tb[1] - tc[1], tb[2] - tc[2] };
                        tn[0] = e1[1] * e2[2] - e1[2] * e2[1];
                        tn[1] = e1[2] * e2[0] - e1[0] * e2[2];
                        tn[2] = e1[0] * e2[1] - e1[1] * e2[0];
                    }
                } else if (nrm.src_n) { tn[0] = nrm.tgt_n[3ll * bidx]; tn[1] = nrm.tgt_n[3ll * bidx + 1]; tn[2] = nrm.tgt_n[3ll * bidx + 2]; }
                float vbx, vby, vbz;
                double dist;
                if (pair_eval(st, p[0], p[1], p[2], qx, qy, qz, nrm, i, tn, thresh, vbx, vby, vbz, dist)) {
                    const double a0 = (double)p4.x - pvx, a1 = (double)p4.y - pvy, a2 = (double)p4.z - pvz;
                    const double b0 = (double)vbx - pvx, b1 = (double)vby - pvy, b2 = (double)vbz - pvz;
                    const double dd = dist - d_pivot;
                    // the pair's terms are wave-uniform: every lane forms them, lane k keeps number k
                    double mine = 0.0;
#define OA_LANE_ADD(k, expr) mine = (lane == (k)) ? (expr) : mine
                    OA_LANE_ADD(S_A, a0); OA_LANE_ADD(S_A + 1, a1); OA_LANE_ADD(S_A + 2, a2);
                    OA_LANE_ADD(S_B, b0); OA_LANE_ADD(S_B + 1, b1); OA_LANE_ADD(S_B + 2, b2);
                    OA_LANE_ADD(S_H + 0, b0 * a0); OA_LANE_ADD(S_H + 1, b0 * a1); OA_LANE_ADD(S_H + 2, b0 * a2);
                    OA_LANE_ADD(S_H + 3, b1 * a0); OA_LANE_ADD(S_H + 4, b1 * a1); OA_LANE_ADD(S_H + 5, b1 * a2);
                    OA_LANE_ADD(S_H + 6, b2 * a0); OA_LANE_ADD(S_H + 7, b2 * a1); OA_LANE_ADD(S_H + 8, b2 * a2);
                    OA_LANE_ADD(S_AA, (a0 * a0 + a1 * a1) + a2 * a2);
                    OA_LANE_ADD(S_BB, (b0 * b0 + b1 * b1) + b2 * b2);
                    OA_LANE_ADD(S_K, 1.0);
                    OA_LANE_ADD(S_D, dd);
                    OA_LANE_ADD(S_DD, dd * dd);
#undef OA_LANE_ADD
                    lane_sum += mine;
                }
            }
        } else if (lane == 0) {
            keys[i] = ((unsigned long long)__float_as_uint(best) << 32) | bidx;
            if (!TRI) win[i] = make_float4(bx, by, bz, __int_as_float((int)bidx));
        }
    }
    if (ACC) {
        // the waves' sums, added in order
        if (lane < NSUMS) red[w][lane] = lane_sum;
        __syncthreads();
        if (threadIdx.x < NSUMS) {
            double t = red[0][threadIdx.x];
            for (int k = 1; k < WPB; ++k) t += red[k][threadIdx.x];
            partials[(long long)blockIdx.x * NSUMS + threadIdx.x] = t;
        }
    }
}

#endif  // __HIPCC__
}  // namespace oa
