// oa_sorted16.hpp -- k_nn_search_sorted16: k_nn_search_sorted with its level 0 in packed half precision (round 5).
//
// EXPERIMENT, NOT BUILT INTO liboa_icp.so.  Kept as the record of a negative result (tools/experiments/README.md): wired in
// (include after oa_kernels.hpp, launch instead of k_nn_search_sorted<R, 256>) it answered the sorted-kernel tests it was run on
// and executed 1.20 VALU instructions per pair instead of 1.84, but took 37.2 - 38.5 ms per 1M <-> 1M search against 35.6 ms:
// in this instruction mix the packed-half instructions cost ~2.05 ns each (1.75 ns alone), the per-tile thresholds and the
// register pressure of the rare path ate the rest.  The best this level could reach (level0_half_loop.hip: 1.77 ns per pair in
// isolation against ~2.0 for the float level) did not justify a second conservative-rounding proof in the product.
//
// WHY.  On gfx950 v_pk_fma_f16 and v_pk_minimum3_f16 issue at half rate for TWO lanes' worth of work each
// (tools/valu_rates.hip, profiles/r05q_valu_rates.txt): one fma per pair costs 0.87 ns per wave-instruction-pair where
// v_sub_f32 costs 0.95, and one comparison costs 0.43 where v_min3_f32 costs 0.87.  Level 0 of the sorted kernel drops from
// ~2.0 to ~1.4 ns per pair and SIMD.  Every (source, target) pair still gets its own arithmetic.
//
// WHAT.  Level 0 is the 1-D score of the pair in TILE-LOCAL, scaled coordinates, in half precision:
//     per tile t of 1024 sorted vertices: origin o_t (the middle of the tile's slab in u), scale S_t = 2^k with
//         (half width of the slab) S_t in (1/16, 1/8]            (k_pack_sorted16; the half width is floored at extent / 8192)
//     per vertex j:   x16_j = fl16((qu_j - o_t) S_t)     stored as  c_j = -2 x16_j (exact)  and  w_j = fl16(x16_j^2)
//     per point, per tile:  y16 = fl16((hu - o_t) S_t)    and a threshold T16 (below)
//     per pair:       s16 = fl16(fma(c_j, y16, w_j))  ~  (x16_j - y16)^2 - y16^2          ONE v_pk_fma_f16 for two pairs
//     a block of 16 vertices is skipped for a point when  min_j s16 > T16                  v_pk_minimum3_f16: four comparisons
// Two points of a lane share the halves of a register (the vertex' c_j, w_j are splat by op_sel): the minimum and the
// threshold stay per point.  Values near the point are small numbers around the tile's origin, so half precision resolves
// them to 2^-11 RELATIVE to their distance from the slab; far slabs overflow harmlessly (+inf > T16: skipped, as they must be).
//
// PROOF that a skipped pair is a loser.  Notation: eps = 2^-11, eta = 2^-14 (the smallest normal half: every bound below
// holds whether the hardware keeps or flushes half denormals), reals x = (qu_j - o) S, y = (hu - o) S, rs = sqrt(base) S
// with base the right-hand side of filter_thresholds (a pair whose REAL centred distance has |q - h|^2 > base is a proven
// loser, docs/HISTORY.md 4.1; the distance along one axis never exceeds it, and S is a power of two).
//   (1) stored position:  |x16 - x| <= eps |x| + eta      (float then half rounding of an exactly scaled difference; x16 whose
//       doubling would be a denormal is stored as 0: an error below eta).  X = max |x16| over the tile, exactly, in the header.
//   (2) the point:        yf = fl32(fl32(hu - o) S), y16 = fl16(yf):  |y16 - y| <= (eps + 2^-23)|y| + eta.  Y = |y16|.
//   (3) the score:        c = -2 x16 exactly, |w - x16^2| <= eps x16^2 + eta, one rounding of the exact fma:
//                         |s16 - A| <= eps (2.01 X^2 + 2 X Y) + 4 eta =: E1,   A = x16^2 - 2 x16 y16 = (x16 - y16)^2 - y16^2
//                         (an overflow to +inf only happens when the exact value is >= 65520 > any finite T16).
//   (4) s16 > T16 and T16 >= R^2 - Y^2 + E1  =>  A > R^2 - Y^2  =>  |x16 - y16| > R.
//   (5) |x - y| >= |x16 - y16| - |x16 - x| - |y16 - y| >= |x16 - y16| - D,  D = 1.01 eps (X + Y) + 3 eta.
//   With R = (rs + D)(1 + 1e-6):  |x - y| > rs, i.e. (qu_j - hu)^2 > base.                                                  qed
//   T16 is R^2 - Y^2 + E1 (+ a float-evaluation slack) rounded UP to half, never a positive denormal, clamped to -65504
//   from below (any threshold above the required one is valid) and +inf from above (nothing is skipped).
//   Far tiles (|yf| > 32768, where y16 would lose the relation to y): if |yf| (1 - 1e-6) - X > rs (1 + 1e-6) every vertex of
//   the tile is beyond rs: y16 = 0, T16 = -65504 (s16 = w_j >= 0 > T16: all skipped); else T16 = +inf (nothing skipped).
//   Tiles whose header could not be formed (no finite slab) and unseeded points (best = inf): T16 = +inf.
// Levels 1 .. 3 (float, exact metric, original indices) are k_nn_search_sorted's: whatever level 0 lets through is decided
// there, so the answers are bit-identical to every other search of the library -- level 0 can only cost time.
#pragma once

namespace oa {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr float S16_EPS = 4.95e-4f;          // >= 1.01 x 2^-11
constexpr float S16_ETA = 6.2e-5f;           // >= 2^-14
constexpr float S16_YMAX = 32768.0f;
constexpr int S16_TG = 256;                  // groups of 4 vertices per tile (1024 vertices): the tile IS the unit of the header

__device__ __forceinline__ uint32_t half_bits(_Float16 h) { return (uint32_t)__builtin_bit_cast(unsigned short, h); }

// Tile images of the sorted target, tile-major: per tile 4 x 256 float4 = 16 KiB,
//   [g]              (c_{4g} c_{4g+1} | c_{4g+2} c_{4g+3} | w_{4g} w_{4g+1} | w_{4g+2} w_{4g+3})  as half pairs   level 0
//   [256 + 3 g ...]  (-2 qu x4) (-2 qv x4) (qu^2 + qv^2 x4)                                        as floats       level 1
// and the header thdr[tile] = (o, S, X, usable ? 1 : 0).  One workgroup of 256 threads per tile.
__global__ __launch_bounds__(256) void k_pack_sorted16(const float *__restrict__ xyz, int nt, const int *__restrict__ order, float cx,
                                                       float cy, float cz, int au, int av, float w_floor, float4 *__restrict__ img,
                                                       float4 *__restrict__ thdr)
{
    __shared__ float red_mn[4], red_mx[4], red_x[4];
    const int t = blockIdx.x, g = threadIdx.x;
    const long long G = (long long)t * S16_TG + g;
    float qu[4], qv[4];
    bool valid[4];
    float mn = INFINITY, mx = -INFINITY;
    for (int k = 0; k < 4; ++k) {
        const long long j = 4ll * G + k;
        valid[k] = j < nt;
        qu[k] = qv[k] = 0.f;
        if (valid[k]) {
            const int v = order[j];
            float q[3];
            q[0] = (float)((double)xyz[3ll * v] - (double)cx);
            q[1] = (float)((double)xyz[3ll * v + 1] - (double)cy);
            q[2] = (float)((double)xyz[3ll * v + 2] - (double)cz);
            qu[k] = q[au]; qv[k] = q[av];
            mn = fminf(mn, qu[k]); mx = fmaxf(mx, qu[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_xor(mn, off, 64)); mx = fmaxf(mx, __shfl_xor(mx, off, 64)); }
    if ((g & 63) == 0) { red_mn[g >> 6] = mn; red_mx[g >> 6] = mx; }
    __syncthreads();
    mn = fminf(fminf(red_mn[0], red_mn[1]), fminf(red_mn[2], red_mn[3]));
    mx = fmaxf(fmaxf(red_mx[0], red_mx[1]), fmaxf(red_mx[2], red_mx[3]));
    // header: origin in the middle of the slab, a power-of-two scale that brings its half width into (1/16, 1/8]
    const float o = (float)(0.5 * ((double)mn + (double)mx));
    const double W = fmax((double)mx - (double)o, (double)o - (double)mn);
    const double We = fmax(W, (double)w_floor);
    bool ok = (mn <= mx) && (o - o == 0.f) && We > 0.0 && We < 1e30;
    float S = 1.f;
    if (ok) {
        int e;
        frexp(0.125 / We, &e);                                    // 0.125 / We = f 2^e, f in [0.5, 1)  ->  2^(e-1) We in (1/16, 1/8]
        ok = (e - 1 >= -100 && e - 1 <= 100);
        if (ok) S = ldexpf(1.f, e - 1);
    }
    _Float16 x16[4];
    float ax = 0.f;
    for (int k = 0; k < 4; ++k) {
        const float xf = ok && valid[k] ? (float)(((double)qu[k] - (double)o) * (double)S) : 0.f;
        _Float16 h = (_Float16)xf;
        if (fabsf((float)h) < 3.1e-5f) h = (_Float16)0.f;         // (below 2^-15 the doubling c = -2 x16 would be a denormal: store 0)
        x16[k] = h;
        ax = fmaxf(ax, fabsf((float)h));
    }
    for (int off = 32; off > 0; off >>= 1) ax = fmaxf(ax, __shfl_xor(ax, off, 64));
    if ((g & 63) == 0) red_x[g >> 6] = ax;
    __syncthreads();
    ax = fmaxf(fmaxf(red_x[0], red_x[1]), fmaxf(red_x[2], red_x[3]));
    ok = ok && ax <= 0.13f;                                      // (cannot fail: |x| <= 1/8 before rounding)
    if (g == 0) thdr[t] = make_float4(o, S, ax, ok ? 1.f : 0.f);

    uint32_t cb[4], wb[4];
    float AU[4], AV[4], W2[4];
    for (int k = 0; k < 4; ++k) {
        if (valid[k]) {
            const float xf = (float)x16[k];
            cb[k] = half_bits((_Float16)(-2.0f * xf));            // exact
            wb[k] = half_bits((_Float16)(xf * xf));               // the float product is exact (2 x 11 bits); one rounding to half
            AU[k] = -2.0f * qu[k]; AV[k] = -2.0f * qv[k];
            W2[k] = (float)((double)qu[k] * (double)qu[k] + (double)qv[k] * (double)qv[k]);
        } else {                                                  // padding can never pass a level, nor win
            cb[k] = 0u; wb[k] = 0x7C00u;                          // w = +inf
            AU[k] = AV[k] = 0.f; W2[k] = 3.0e38f;
        }
    }
    float4 *dst = img + (long long)t * (4 * S16_TG);
    dst[g] = make_float4(__uint_as_float(cb[0] | (cb[1] << 16)), __uint_as_float(cb[2] | (cb[3] << 16)),
                         __uint_as_float(wb[0] | (wb[1] << 16)), __uint_as_float(wb[2] | (wb[3] << 16)));
    dst[S16_TG + 3 * g] = make_float4(AU[0], AU[1], AU[2], AU[3]);
    dst[S16_TG + 3 * g + 1] = make_float4(AV[0], AV[1], AV[2], AV[3]);
    dst[S16_TG + 3 * g + 2] = make_float4(W2[0], W2[1], W2[2], W2[3]);
}

// y16 and T16 of one point for one tile (header H = (o, S, X, usable)); rho = round_up(sqrt(base) (1 + 2^-22)) is the point's
// level-0 radius of k_nn_search_sorted (sorted_thresholds' thr1).  See the proof at the top of this file.
__device__ __forceinline__ void sorted16_setup(float hu, float rho, const float4 H, _Float16 &y16, _Float16 &T16)
{
    y16 = (_Float16)0.f;
    T16 = (_Float16)INFINITY;                                     // nothing is skipped unless proven below
    if (!(H.w > 0.f) || !(rho < INFINITY)) return;
    const float S = H.y, X = H.z;
    const float yf = (hu - H.x) * S, rs = rho * S;
    const float ay = fabsf(yf);
    if (!(ay <= S16_YMAX)) {                                      // a far tile (or a NaN: nothing skipped)
        if (ay * 0.999999f - X > rs * 1.000001f) T16 = (_Float16)(-65504.f);
        return;
    }
    y16 = (_Float16)yf;
    const float Y = fabsf((float)y16);
    const float D = S16_EPS * (X + Y) + 3.f * S16_ETA;
    const float Rr = (rs + D) * 1.000001f;
    const float E1 = S16_EPS * (2.01f * X * X + 2.f * X * Y) + 4.f * S16_ETA;
    const float YY = Y * Y;                                       // exact: 11 x 11 bits
    const float RR = Rr * Rr;
    const float Tf = (__builtin_fmaf(Rr, Rr, -YY) + E1) + 2e-6f * (RR + YY);
    if (!(Tf <= 65504.f)) return;                                 // (also NaN) too wide for half: nothing skipped
    if (Tf < -65504.f) { T16 = (_Float16)(-65504.f); return; }
    if (Tf > 0.f && Tf < 6.2e-5f) { T16 = (_Float16)6.2e-5f; return; }       // never a positive denormal (6.2e-5 rounds to >= 2^-14)
    _Float16 h = (_Float16)Tf;
    if ((float)h < Tf) {                                          // next half up
        unsigned short b = __builtin_bit_cast(unsigned short, h);
        if ((float)h > 0.f) b += 1;
        else if ((float)h < 0.f) b -= 1;
        else b = 0x0400;                                          // above zero: the smallest normal
        h = __builtin_bit_cast(_Float16, b);
    }
    T16 = h;
}

template <int R>
__global__ __launch_bounds__(NN_THREADS, (R <= 4 ? 4 : 2)) void k_nn_search_sorted16(const DevState *__restrict__ st,
                                                                   const float4 *__restrict__ src4,
                                                                   const float4 *__restrict__ tgs,
                                                                   const float4 *__restrict__ img,
                                                                   const float4 *__restrict__ thdr,
                                                                   const float4 *__restrict__ tf3s,
                                                                   const int4 *__restrict__ tidx,
                                                                   const float4 *__restrict__ win,
                                                                   int n_groups_pad, int au, int av,
                                                                   unsigned long long *__restrict__ keys)
{
    static_assert(R % 2 == 0, "k_nn_search_sorted16: two points share the halves of a register");
    if (st->halt) return;
    constexpr int TG = S16_TG, TILE_F4 = 4 * TG, LOADS = TILE_F4 / NN_THREADS;
    static_assert(LOADS == 4, "k_nn_search_sorted16: 4 float4 per thread and tile");
    __shared__ float4 tile[2][TILE_F4];
    __shared__ short ord[SORT_ORDER_MAX];
    const int tid = threadIdx.x;
    const double qmax = st->qmax;
    const float cx = st->tc[0], cy = st->tc[1], cz = st->tc[2];
    const int base = blockIdx.y * (NN_THREADS * R);
    const int wave_base = (tid >> 6) * (64 * R) + (tid & 63);     // a wave owns R x 64 consecutive slots (k_nn_search_sorted)
#define OA_SLOT(r) (base + wave_base + (r) * 64)
    float px[R], py[R], pz[R], hu[R], hv[R], hd[R], best[R], rho[R], thr2[R];
    uint32_t bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = OA_SLOT(r);
        const float4 p = src4[i];
        co_find(st, p.x, p.y, p.z, px[r], py[r], pz[r]);    // co_find (general.py:287)
        const float h0 = (float)((double)px[r] - (double)cx);
        const float h1 = (float)((double)py[r] - (double)cy);
        const float h2 = (float)((double)pz[r] - (double)cz);
        hu[r] = au == 0 ? h0 : (au == 1 ? h1 : h2);
        hv[r] = av == 0 ? h0 : (av == 1 ? h1 : h2);
        hd[r] = (au + av == 1) ? h2 : ((au + av == 2) ? h1 : h0);
        best[r] = INFINITY;
        bidx[r] = IDX_NONE;
        const float4 sw = win ? win[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (__float_as_int(sw.w) >= 0) {
            const float d = d2_metric(px[r], py[r], pz[r], sw.x, sw.y, sw.z);
            if (d < INFINITY) { best[r] = d; bidx[r] = (uint32_t)__float_as_int(sw.w); }
        }
        sorted_thresholds(best[r], hu[r], hv[r], hd[r], qmax, rho[r], thr2[r]);
    }

    int g_begin, g_end;
    split_range(n_groups_pad, TG, g_begin, g_end);
    const int n_tiles = (g_end - g_begin) / TG;
    const int tile0 = g_begin / TG;                                // this split's first tile
    const float4 *tsrc = img + (long long)tile0 * TILE_F4;

    // visiting order of this split's tiles: middle-out from the slab nearest (in u) to the workgroup's first point
    const bool ordered = n_tiles <= SORT_ORDER_MAX;
    if (ordered && tid == 0) {
        int a = 0, b = n_tiles;                                    // tiles [0, a): slab origin at or before the point
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (thdr[tile0 + mid].x <= hu[0]) a = mid + 1; else b = mid;
        }
        const int s = a > 0 ? a - 1 : 0;
        int lo = s - 1, hi = s + 1, n = 0;
        ord[n++] = (short)s;
        while (n < n_tiles) {
            if (hi < n_tiles) ord[n++] = (short)hi++;
            if (lo >= 0 && n < n_tiles) ord[n++] = (short)lo--;
        }
    }
    __syncthreads();
#define OA_TILE_AT(k) (ordered ? (int)ord[(k)] : (k))

    float4 stg0, stg1, stg2, stg3, Hn;
    {
        const int t0 = OA_TILE_AT(0);
        const float4 *fsrc = tsrc + (long long)TILE_F4 * t0;
        stg0 = fsrc[tid]; stg1 = fsrc[NN_THREADS + tid]; stg2 = fsrc[2 * NN_THREADS + tid]; stg3 = fsrc[3 * NN_THREADS + tid];
        tile[0][tid] = stg0; tile[0][NN_THREADS + tid] = stg1; tile[0][2 * NN_THREADS + tid] = stg2; tile[0][3 * NN_THREADS + tid] = stg3;
        Hn = thdr[tile0 + t0];
    }
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < n_tiles);
        const float4 H = Hn;
        const int tcur = OA_TILE_AT(t);
        if (more) {                                               // next tile: global -> registers, hidden under compute
            const int tn = OA_TILE_AT(t + 1);
            const float4 *nsrc = tsrc + (long long)TILE_F4 * tn;
            stg0 = nsrc[tid]; stg1 = nsrc[NN_THREADS + tid]; stg2 = nsrc[2 * NN_THREADS + tid]; stg3 = nsrc[3 * NN_THREADS + tid];
            Hn = thdr[tile0 + tn];
        }
        // this tile's y16 / T16 of the lane's points: points 2p and 2p + 1 in the halves of one register
        half2v Y[R / 2], T[R / 2];
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
            _Float16 y0, t0, y1, t1;
            sorted16_setup(hu[2 * p], rho[2 * p], H, y0, t0);
            sorted16_setup(hu[2 * p + 1], rho[2 * p + 1], H, y1, t1);
            Y[p] = half2v{y0, y1};
            T[p] = half2v{t0, t1};
        }
        const long long gbase = g_begin + (long long)tcur * TG;
        constexpr int GW = 4;                                     // 16 vertices x R points per skip test
        for (int g = 0; g < TG; g += GW) {
            float4 P[GW];
#pragma unroll
            for (int k = 0; k < GW; ++k) P[k] = tile[cur][g + k];
            bool hit0 = false, hit[R];
#pragma unroll
            for (int p = 0; p < R / 2; ++p) {                      // level 0: half a v_pk_fma_f16 + a quarter v_pk_minimum3_f16 per pair
                half2v s[4 * GW];
#pragma unroll
                for (int k = 0; k < GW; ++k) {
                    const half2v C01 = __builtin_bit_cast(half2v, P[k].x), C23 = __builtin_bit_cast(half2v, P[k].y);
                    const half2v W01 = __builtin_bit_cast(half2v, P[k].z), W23 = __builtin_bit_cast(half2v, P[k].w);
                    s[4 * k] = __builtin_elementwise_fma(half2v{C01.x, C01.x}, Y[p], half2v{W01.x, W01.x});
                    s[4 * k + 1] = __builtin_elementwise_fma(half2v{C01.y, C01.y}, Y[p], half2v{W01.y, W01.y});
                    s[4 * k + 2] = __builtin_elementwise_fma(half2v{C23.x, C23.x}, Y[p], half2v{W23.x, W23.x});
                    s[4 * k + 3] = __builtin_elementwise_fma(half2v{C23.y, C23.y}, Y[p], half2v{W23.y, W23.y});
                }
                half2v m = s[0];
#pragma unroll
                for (int k = 1; k + 1 < 4 * GW; k += 2) m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m, s[k]), s[k + 1]);
                m = __builtin_elementwise_minimum(m, s[4 * GW - 1]);
                hit[2 * p] = !(m.x > T[p].x);
                hit[2 * p + 1] = !(m.y > T[p].y);
                hit0 |= hit[2 * p] | hit[2 * p + 1];
            }
            if (!hit0) continue;
#pragma unroll 1
            for (int k = 0; k < GW; ++k) {                         // rare from here on: one group of 4 vertices at a time, in float
                const float4 AU = tile[cur][TG + 3 * (g + k)], AV = tile[cur][TG + 3 * (g + k) + 1], W2 = tile[cur][TG + 3 * (g + k) + 2];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!hit[r]) continue;
                    if (sorted_finish_group(AU, AV, W2, gbase + g + k, tf3s, tgs, tidx, px[r], py[r], pz[r], hu[r], hv[r], hd[r], qmax,
                                            best[r], bidx[r], rho[r], thr2[r])) {
                        _Float16 yn, tn;                           // a tighter radius: this tile's threshold again
                        sorted16_setup(hu[r], rho[r], H, yn, tn);
                        if (r & 1) { Y[r / 2].y = yn; T[r / 2].y = tn; } else { Y[r / 2].x = yn; T[r / 2].x = tn; }
                    }
                }
            }
        }
        if (more) {
            tile[cur ^ 1][tid] = stg0; tile[cur ^ 1][NN_THREADS + tid] = stg1;
            tile[cur ^ 1][2 * NN_THREADS + tid] = stg2; tile[cur ^ 1][3 * NN_THREADS + tid] = stg3;
        }
        __syncthreads();
    }
#undef OA_TILE_AT

    // a split reports when it has something to say (k_nn_search_filtered); the seed's owner: seed index mod splits
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
        unsigned long long *dst = keys + OA_SLOT(r);
        if (gridDim.x == 1) *dst = key;
        else {
            uint32_t seed_idx = IDX_NONE;                          // the seed again, from the slot's record (k_nn_search_sorted)
            float seed_d = INFINITY;
            const float4 sw = win ? win[OA_SLOT(r)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (__float_as_int(sw.w) >= 0) {
                const float d = d2_metric(px[r], py[r], pz[r], sw.x, sw.y, sw.z);
                if (d < INFINITY) { seed_d = d; seed_idx = (uint32_t)__float_as_int(sw.w); }
            }
            const bool seeded = seed_idx != IDX_NONE;
            const bool improved = bidx[r] != seed_idx || best[r] != seed_d;
            const bool owner = seeded && (seed_idx % gridDim.x) == blockIdx.x;
            if (!seeded || improved || owner) atomicMin(dst, key);
        }
    }
#undef OA_SLOT
}

}  // namespace oa
