// oa_mfma.hpp -- EXPERIMENT (OA_NN_MFMA=1, off by default): the first filter level of the brute-force nearest-vertex
// search on the matrix cores.
//
// BASELINE.json's north-star describes the brute-force search without MFMA, and the default kernel
// (k_nn_search_filtered, oa_kernels.hpp) honours that: it sits at the fp32 vector issue limit with 2 FMA + 1 min slot per
// pair.  VERDICT r1 item 8 asked what a low-precision MFMA pre-filter would buy; this file is the answer, behind a flag.
//
// Same structure as k_nn_search_filtered (a conservative score that can only PROVE LOSERS; everything it cannot rule out
// goes through that kernel's level 3: the exact metric, lexicographic (d2, index)), with levels 1 and 2 replaced: one
// v_mfma_f32_32x32x16_f16 scores 32 targets x 32 points -- 1024 pairs -- and the sign of each of its results says
// "cannot win or tie" (+) or "look closer" (-).
//
//   scaled centred coordinates   x = sigma * fl32(q - c) (targets, |x| <= 1),  y = sigma * fl32(p - c) (points),  sigma = 2^e
//   target row   U = [ah0 al0 ah0 | ah1 al1 ah1 | ah2 al2 ah2 | wh wm wl | -1 -1 -1 | 0]     a = -2x = ah + al (+ r_a),
//   point column V = [ph0 ph0 pl0 | ph1 ph1 pl1 | ph2 ph2 pl2 |  1  1  1 | th tm tl | 0]     y = ph + pl (+ r_y),
//                                                                 w = sigma^2 |q^|^2 = wh + wm + wl (+ r_w)   (binary16 parts)
//   U . V = sum_c a_c y_c + w - T'  up to the dropped al*pl and residual terms  =  sigma^2 S_j - T'  (+ error),
//           S_j = |q^_j|^2 - 2 p^ . q^_j   (the 3-D score of k_nn_search_filtered, exact arithmetic)
//   T' = sigma^2 (best (1 + 16u) + 16u G^2 + 1e-30 - |p^|^2) + E,   th + tm + tl >= T' (last part rounded up)
//
// Error budget E (scaled units, u = 2^-24, Y = max_c |y_c| <= 64):
//   dropped al*pl: |al| <= 2^-11, |pl| <= 2^-11 Y  -> 2^-22 Y per axis;  r_a <= 2^-22, times |y| <= Y;  r_y <= 2^-22 Y,
//   times |a| <= 2: together 2^-20 Y per axis = 48u Y over the three axes; binary16 underflow (parts below 2^-24 are
//   lost): <= 9u in all; the matrix core's own arithmetic: measured <= 4.2u sum|U_k V_k| on adversarial data
//   (tools/mfma_microbench.hip, profiles/r02r_mfma_microbench.txt), budgeted 16u M with M >= sum|U_k V_k|.
//   E = u (48 Y + 9 + 16 M),  M = 6.02 Y + 3.02 + 1.01 |sigma^2 thr| + 1.
// Claim: result >= +0 (sign bit clear)  =>  sigma^2 S_j >= T' - E  =>  S_j + |p^|^2 >= best (1 + 16u) + 16u G^2, which is
// more than k_nn_search_filtered's proof needs (11.99u G^2) to conclude d2_metric(p, q_j) > best.  Columns that cannot be
// represented (no best yet, Y > 64, |T'| > 16384, non-finite) are set to "always look closer" (result -1).
#pragma once
#include "oa_kernels.hpp"

namespace oa {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int MF_TILES = 16;                  // MFMA tiles (32 targets, 1 KiB) per LDS buffer
constexpr int MF_GROUPS = MF_TILES * 8;       // = 128 groups of 4 targets

#if defined(__HIPCC__)

// x = h + l + r with |r| <= 2^-11 |l| (or binary16 underflow)
__device__ __forceinline__ void split2_f16(double x, _Float16 &h, _Float16 &l)
{
    h = (_Float16)x;
    l = (_Float16)(x - (double)h);
}

// MFMA image of the target: per tile of 32 targets 64 x half8 -- slot (plane k8, row i) at tile * 64 + k8 * 32 + i holds
// U[8 k8 .. 8 k8 + 7] of target 32 tile + i, so lane l of a wave reads slot tile * 64 + l (one contiguous KiB per tile).
// Padding targets get w = 60000 (never below any representable threshold).
#if !defined(OA_FAMILY_TU) && defined(OA_EXPERIMENTS)      // experiment: only in liboa_icp_exp.so's host translation unit
__global__ void k_pack_filter_mfma(const float *__restrict__ xyz, int nt, int n_targets_pad, float cx, float cy, float cz,
                                   int au, int av, int ad, double sigma, half8 *__restrict__ img)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_targets_pad) return;
    _Float16 U[16];
    for (int k = 0; k < 16; ++k) U[k] = (_Float16)0.f;
    if (j < nt) {
        float q[3];
        q[0] = (float)((double)xyz[3ll * j] - (double)cx);
        q[1] = (float)((double)xyz[3ll * j + 1] - (double)cy);
        q[2] = (float)((double)xyz[3ll * j + 2] - (double)cz);
        const int ax[3] = { au, av, ad };
        double w = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double x = (double)q[ax[c]] * sigma;              // exact: sigma is a power of two
            _Float16 h, l;
            split2_f16(-2.0 * x, h, l);
            U[3 * c] = h; U[3 * c + 1] = l; U[3 * c + 2] = h;
            w += x * x;
        }
        const _Float16 wh = (_Float16)w;
        const _Float16 wm = (_Float16)(w - (double)wh);
        const _Float16 wl = (_Float16)(w - (double)wh - (double)wm);
        U[9] = wh; U[10] = wm; U[11] = wl;
    } else U[9] = (_Float16)60000.f;
    U[12] = U[13] = U[14] = (_Float16)(-1.f);
    const int tile = j >> 5, i = j & 31;
    half8 lo, hi;
    for (int k = 0; k < 8; ++k) { lo[k] = U[k]; hi[k] = U[8 + k]; }
    img[(long long)tile * 64 + i] = lo;
    img[(long long)tile * 64 + 32 + i] = hi;
}
#endif  // !OA_FAMILY_TU

// the column of one point: v0 = V[0..7], v1 = V[8..15]
__device__ __forceinline__ void mfma_point_column(float hu, float hv, float hd, float best, double qmax, double sigma,
                                                  half8 &v0, half8 &v1)
{
    const _Float16 z = (_Float16)0.f, one = (_Float16)1.f;
    v0 = half8{ z, z, z, z, z, z, z, z };
    v1 = half8{ z, z, z, z, one, z, z, z };                         // "always look closer": U . V = -1
    if (!(best < INFINITY)) return;
    const double y[3] = { (double)hu * sigma, (double)hv * sigma, (double)hd * sigma };
    const double Y = fmax(fabs(y[0]), fmax(fabs(y[1]), fabs(y[2])));
    if (!(Y <= 64.0)) return;
    const double P3 = (double)hu * (double)hu + (double)hv * (double)hv + (double)hd * (double)hd;
    const double G = sqrt(P3) * (1.0 + 1e-12) + qmax;
    const double thr = ((double)best * (1.0 + FILTER_K) + FILTER_K * G * G + FILTER_ABS - P3) * sigma * sigma;
    const double u = 5.9604644775390625e-08;
    const double M = 6.02 * Y + 3.02 + 1.01 * fabs(thr) + 1.0;
    const double T = thr + u * (48.0 * Y + 9.0 + 16.0 * M);
    if (!(fabs(T) <= 16384.0)) return;
    _Float16 ph[3], pl[3];
    for (int c = 0; c < 3; ++c) split2_f16(y[c], ph[c], pl[c]);
    const _Float16 th = (_Float16)T;
    const _Float16 tm = (_Float16)(T - (double)th);
    const double rem = T - (double)th - (double)tm;
    _Float16 tl = (_Float16)rem;
    if ((double)tl < rem) {                                         // round the last part up: th + tm + tl >= T
        unsigned short b = __builtin_bit_cast(unsigned short, tl);
        if (tl > z) b += 1; else if (tl < z) b -= 1; else b = 1;
        tl = __builtin_bit_cast(_Float16, b);
    }
    v0 = half8{ ph[0], ph[0], pl[0], ph[1], ph[1], pl[1], ph[2], ph[2] };
    v1 = half8{ pl[2], one, one, one, th, tm, tl, z };
}

__device__ __forceinline__ int mfma_sign_or(const float16v d)
{
    int r = __float_as_int(d[0]) | __float_as_int(d[1]) | __float_as_int(d[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) r = r | __float_as_int(d[k]) | __float_as_int(d[k + 1]);
    return r | __float_as_int(d[15]);
}

__device__ __forceinline__ half8 shfl_xor32_half8(half8 v)
{
    typedef int int4_ __attribute__((ext_vector_type(4)));
    int4_ w = __builtin_bit_cast(int4_, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __shfl_xor(w[k], 32, 64);
    return __builtin_bit_cast(half8, w);
}

// Same launch geometry and the same reporting as k_nn_search_filtered<4, 256>; `tfm` is the MFMA image (through LDS),
// `tg` the exact target image (read from global memory by the slow path).  A wave's 256 points form 8 blocks of 32 columns: block
// (r, h) = register r of lanes 32 h .. 32 h + 31.
template <int WPS>
__global__ __launch_bounds__(NN_THREADS, WPS) void k_nn_search_mfma(const DevState *__restrict__ st,
                                                                const float4 *__restrict__ src4,
                                                                const float4 *__restrict__ tg,
                                                                const half8 *__restrict__ tfm,
                                                                const float4 *__restrict__ win,
                                                                int n_groups_pad, double sigma,
                                                                unsigned long long *__restrict__ keys)
{
    constexpr int R = 4;
    if (st->halt) return;
    __shared__ half8 tile[2][MF_TILES * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const double qmax = st->qmax;
    const float cx = st->tc[0], cy = st->tc[1], cz = st->tc[2];
    const int au = st->fax[0], av = st->fax[1];
    const int base = blockIdx.y * (NN_THREADS * R);
    // (the point itself and its seed are re-read where they are needed -- the slow path and the report -- instead of
    //  living in registers through the hot loop: 8 column blocks x 4 VGPRs + 16 results leave no room for them)
    float hu[R], hv[R], hd[R], best[R];
    uint32_t bidx[R];
    half8 Bf[2 * R];
#define OA_MF_POINT(r, X, Y, Z)                                                                                     \
    float X, Y, Z;                                                                                                   \
    {                                                                                                                \
        const float4 p_ = src4[base + (r) * NN_THREADS + tid];                                                       \
        co_find(st, p_.x, p_.y, p_.z, X, Y, Z);                /* co_find (general.py:287) */                   \
    }
#define OA_MF_SEED(r, X, Y, Z, SD, SI)                                                                              \
    float SD = INFINITY;                                                                                             \
    uint32_t SI = IDX_NONE;                                                                                          \
    {                                                                                                                \
        const float4 sw_ = win ? win[base + (r) * NN_THREADS + tid] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1)); \
        if (__float_as_int(sw_.w) >= 0) {                                                                            \
            const float d_ = d2_metric(X, Y, Z, sw_.x, sw_.y, sw_.z);                                                \
            if (d_ < INFINITY) { SD = d_; SI = (uint32_t)__float_as_int(sw_.w); }                                    \
        }                                                                                                            \
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        OA_MF_POINT(r, qx, qy, qz);
        const float h0 = (float)((double)qx - (double)cx);
        const float h1 = (float)((double)qy - (double)cy);
        const float h2 = (float)((double)qz - (double)cz);
        hu[r] = au == 0 ? h0 : (au == 1 ? h1 : h2);
        hv[r] = av == 0 ? h0 : (av == 1 ? h1 : h2);
        hd[r] = (au + av == 1) ? h2 : ((au + av == 2) ? h1 : h0);
        OA_MF_SEED(r, qx, qy, qz, sd, si);
        best[r] = sd;
        bidx[r] = si;
    }
    // the columns of block (r, 0) belong to lanes 0..31, of block (r, 1) to lanes 32..63; lane l supplies rows
    // k = 8 (l / 32) .. + 7 of column l % 32
#define OA_MF_REBUILD(r)                                                                                             \
    do {                                                                                                             \
        half8 v0, v1;                                                                                                \
        mfma_point_column(hu[r], hv[r], hd[r], best[r], qmax, sigma, v0, v1);                                        \
        const half8 x1 = shfl_xor32_half8(v1), x0 = shfl_xor32_half8(v0);                                            \
        Bf[2 * (r)] = lane < 32 ? v0 : x1;                                                                           \
        Bf[2 * (r) + 1] = lane < 32 ? x0 : v1;                                                                       \
    } while (0)
#pragma unroll
    for (int r = 0; r < R; ++r) OA_MF_REBUILD(r);

    int g_begin, g_end;
    split_range(n_groups_pad, FTILE_GROUPS, g_begin, g_end);      // whole tiles of 256 groups, as k_nn_search_filtered<4, 256>
    const int n_bufs = (g_end - g_begin) / MF_GROUPS;
    const half8 *tsrc = tfm + (long long)(g_begin / 8) * 64;       // 8 groups per MFMA tile
    // Tiles go global -> LDS directly (LDS-DMA, 16 B per lane: the destination is the wave's base + lane x 16, which is
    // exactly this image's layout).  Staging them through registers cost 16 VGPRs the hot loop does not have: the
    // compiler parked them in scratch (PMC: 28 GB of HBM-side writes per launch) and waited for every load at once.
    constexpr int LOADS = MF_TILES * 64 / NN_THREADS;              // 4 x 16 B per thread and buffer
    typedef const void __attribute__((address_space(1))) *gptr_t;
    typedef void __attribute__((address_space(3))) *lptr_t;
#define OA_MF_FETCH(buf, src)                                                                                        \
    do {                                                                                                             \
        _Pragma("unroll") for (int k_ = 0; k_ < LOADS; ++k_)                                                          \
            __builtin_amdgcn_global_load_lds((gptr_t)((src) + k_ * NN_THREADS + tid), (lptr_t)&tile[buf][k_ * NN_THREADS + tid], 16, 0, 0); \
    } while (0)
    OA_MF_FETCH(0, tsrc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float16v zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

    for (int t = 0; t < n_bufs; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < n_bufs);
        if (more) OA_MF_FETCH(cur ^ 1, tsrc + (long long)(t + 1) * MF_TILES * 64);   // lands while this buffer is consumed
        for (int mt = 0; mt < MF_TILES; ++mt) {
            const half8 a = tile[cur][mt * 64 + lane];
            const int g0 = g_begin + t * MF_GROUPS + mt * 8;       // first of the tile's 8 groups
            // Hot path: the eight MFMAs' results are OR-ed into ONE running word (8 x v_or3_b32 per MFMA, nothing else) and
            // the sign is tested once per tile.  v_or3 is a half-rate instruction on this part: eight of them take as long
            // as the MFMA they follow, so each MFMA is issued one step ahead of the ORs that read it.  (Tried, slower: two
            // tiles per trip, 24.5 vs 23.3 ms; s_setprio(1) around the MFMA, 23.8; carrying the pipeline across trips --
            // the compiler then hoists four MFMAs to the top of the loop and spills their results.)
            int acc = 0;
            float16v d_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bf[0], zero, 0, 0, 0);
#pragma unroll
            for (int blk = 0; blk < 2 * R; ++blk) {
                const float16v d = d_next;
                if (blk + 1 < 2 * R) d_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bf[blk + 1], zero, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 16; k += 2) acc = acc | __float_as_int(d[k]) | __float_as_int(d[k + 1]);
            }
            if (!__any(acc < 0)) continue;
            // Slow path (about one tile per point and search): the tile again, block by block
#pragma unroll
            for (int blk = 0; blk < 2 * R; ++blk) {
                const float16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bf[blk], zero, 0, 0, 0);
                const int s = mfma_sign_or(d);
                if (__any(s < 0)) {
                    // some (target, point) of this tile and block could not be ruled out.  The signs say which: result k of
                    // lane l is target row (k & 3) + 8 (k >> 2) + 4 (l >> 5) of the tile, column l & 31.  The owner lane
                    // of a column collects its two halves of the rows and evaluates exactly those targets (level 3 of
                    // k_nn_search_filtered: the exact metric, lexicographic (d2, index)); the MFMA score is as sharp as
                    // that kernel's fp32 level 2, which is skipped.
                    const int r = blk >> 1, h = blk & 1;
                    unsigned m = 0u;
#pragma unroll
                    for (int k = 0; k < 16; ++k) m |= ((unsigned)__float_as_int(d[k]) >> 31) << k;
                    const unsigned mo = (unsigned)__shfl_xor((int)m, 32, 64);
                    bool improved = false;
                    if ((lane >> 5) == h && (m | mo) != 0u) {
                        OA_MF_POINT(r, qx, qy, qz);
                        float b = best[r];
                        uint32_t bi = bidx[r];
                        unsigned long long todo = (unsigned long long)m | ((unsigned long long)mo << 16);   // own rows, then the partner's
                        while (todo) {
                            const int bit = __ffsll((long long)todo) - 1;
                            todo &= todo - 1ull;
                            const int k = bit & 15, half = (bit < 16) ? h : (h ^ 1);
                            const int row = (k & 3) + 8 * (k >> 2) + 4 * half;
                            const long long g = g0 + (row >> 2);
                            const int comp = row & 3;
                            const float4 *eg = tg + 3 * g;
                            const float4 X = eg[0], Yv = eg[1], Z = eg[2];
                            const float tx = comp == 0 ? X.x : (comp == 1 ? X.y : (comp == 2 ? X.z : X.w));
                            const float ty = comp == 0 ? Yv.x : (comp == 1 ? Yv.y : (comp == 2 ? Yv.z : Yv.w));
                            const float tz = comp == 0 ? Z.x : (comp == 1 ? Z.y : (comp == 2 ? Z.z : Z.w));
                            const uint32_t j = (uint32_t)g * 4u + (uint32_t)comp;
                            const float e = d2_metric(qx, qy, qz, tx, ty, tz);
                            if (e < b || (e == b && j < bi)) { b = e; bi = j; }
                        }
                        improved = b < best[r];
                        best[r] = b;
                        bidx[r] = (b < INFINITY) ? bi : IDX_NONE;       // overflowed distances (+inf) never win
                    }
                    if (__any(improved)) OA_MF_REBUILD(r);           // a tighter threshold for the columns that improved
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's part of the next buffer has landed ...
        __syncthreads();                                            // ... and so has everybody else's; nobody still reads this one
    }
#undef OA_MF_FETCH
#undef OA_MF_REBUILD

    const uint32_t own_lo = (uint32_t)g_begin * 4u, own_hi = (uint32_t)g_end * 4u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
        unsigned long long *dst = keys + base + r * NN_THREADS + tid;
        if (gridDim.x == 1) *dst = key;
        else {
            OA_MF_POINT(r, qx, qy, qz);
            OA_MF_SEED(r, qx, qy, qz, sd, si);
            const bool seeded = si != IDX_NONE;
            const bool improved = bidx[r] != si || best[r] != sd;
            const bool owner = seeded && si >= own_lo && si < own_hi;
            if (!seeded || improved || owner) atomicMin(dst, key);  // (d2, idx) lexicographic: lowest index on ties
        }
    }
}

#undef OA_MF_POINT
#undef OA_MF_SEED

#endif  // __HIPCC__
}  // namespace oa
