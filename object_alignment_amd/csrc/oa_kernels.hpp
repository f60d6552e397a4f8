// oa_kernels.hpp -- gfx950 (CDNA4, wave64) device code of the ICP hot path.
//
// One ICP iteration inside the loop (oa_run / oa_iterate) is two launches:
//   search + accumulate   k_nn_search_grid<L, ACC> (oa_grid.hpp), k_bvh_search<TRI, ACC> (oa_bvh.hpp) or, for the modes whose
//                         search has no accumulating form (brute force, triangle grid), the search followed by
//                         k_pair_accumulate_canon: world-space threshold test + fp64 sums (functions/general.py:299-306 fused
//                         with the reductions affine_matrix_from_points needs, :160-167,:181,:208-212), one row of NSUMS
//                         doubles per workgroup, fixed order (bitwise reproducible, no float atomics)
//   k_reduce_solve_update rows -> sums -> 3x3 Kabsch/SVD solve, matrix_world update, convergence ring
//                         (operators/icp_align.py:106-149); across devices k_reduce_post -> k_gather_solve_update (mailboxes)
//                         or k_reduce_partials -> ncclAllReduce -> k_solve_update
// This file: the state (DevState), the float32 "mathutils" arithmetic, the brute-force searches
//   k_nn_search / k_nn_search_filtered   nearest target vertex per source point; target tiles staged through LDS and read
//                         back as wave-uniform (broadcast) ds_read_b128; fp32 VALU bound (docs/HISTORY.md 4.1)
// the pair test and the accumulation helpers (pair_eval, block_store_pair), the row reduction and the solve.
//
// Arithmetic conventions are spelled out in DESIGN.md ("float32 semantics") and are shared bit-for-bit with the
// CPU oracle.  This translation unit is compiled with -ffp-contract=off: every fma below is written explicitly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace oa {

constexpr int NSUMS = 24;
// layout of the per-iteration sums (all relative to `pivot`, a' = a - pivot, b' = b - pivot):
//   [0..2] sum a'   [3..5] sum b'   [6..14] sum b'_i a'_j (row i, col j)   [15] sum |a'|^2   [16] sum |b'|^2
//   [17] K          [18] sum (d - d_pivot)      [19] sum (d - d_pivot)^2        [20..23] reserved (0)
constexpr int S_A = 0, S_B = 3, S_H = 6, S_AA = 15, S_BB = 16, S_K = 17, S_D = 18, S_DD = 19;

constexpr unsigned long long KEY_EMPTY = ~0ull;
constexpr uint32_t IDX_NONE = 0xFFFFFFFFu;

constexpr int NN_THREADS = 256;
constexpr int TILE_GROUPS = 256;          // groups of 4 targets per LDS tile (1024 targets, 12 KiB)
constexpr int ACC_THREADS = 256;
constexpr int ACC_MAX_BLOCKS = 4096;

struct StepRecord {
    double M[16];
    float  new_mat[16];
    double K, mean_d, std_d, trans, angle;
    double search_ticks;      // wall_clock64 ticks from the end of the previous iteration to the start of k_pair_accumulate
    double exchange_ticks;    // ... from "this device's sums are ready" (end of k_reduce_post / k_reduce_partials) to "the world's sums are in
                              // hand" (the gather's wait is over / the solve behind the all-reduce starts); 0 on one GPU
};

struct DevState {
    float  mx1[16], mx2[16], imx1[16], imx2[16];   // align / base matrix_world and inverses (float32, row-major)
    double pivot[3];
    double thresh, target_d;
    double ring_t[5], ring_r[5];
    int32_t iters, use_target, with_scale, early_exit;
    int32_t n, converged, status, halt;
    int32_t max_records, pad0;
    // conservative-filter data (k_nn_search_filtered): target bbox centre and max |q - centre|
    float  tc[3], pad1;
    double qmax;
    int32_t fax[3], pad2;    // filter axes: fax[0], fax[1] kept by the 2-D score, fax[2] (smallest extent) dropped
    double d_pivot;      // subtracted from the pair distances before summing (previous iteration's mean): keeps
                         // the one-pass variance sum d^2 - K mean^2 free of cancellation
    // Search radius (grid / tree searches only): a query whose nearest primitive is farther than
    // cut_a + cut_b (|x| + |y| + |z|) in base-local space is certain to fail `dist < thresh` (general.py:300), so
    // the search may stop there.  cut_a = +inf switches it off (oa_nn_search, brute-force mode).  See search_cutoff().
    double cut_a, cut_b;
    double local_per_world;   // 1 / sigma_min(mx2): upper bound of |local| / |world| distances (0 = unknown)
    int32_t *host_halt;       // pinned host words {halt, n} mirrored for oa_run (stop enqueuing / stay near the GPU), or nullptr
    // GPU-side timing of the search (no hipEvents in the stream: they cost ~3 us each): k_stamp_start / the solve
    // kernel leave the end of the previous iteration in t_prev_end, k_pair_accumulate its own start in t_acc_start
    unsigned long long t_prev_end, t_acc_start;
    unsigned long long t_xchg_start;  // multi-GPU: when this device's sums of the iteration were ready for the exchange (StepRecord::exchange_ticks)
    // shader cycles and constant-rate ticks one workgroup in the middle of the last k_nn_search_sorted launch lived for: the clock
    // the chip held DURING the search (OA_STAT_SEARCH_CLOCK_MHZ; bench.py prices the measured issue rates at this clock)
    unsigned long long search_clk[2];
    // Shards in the zone where the tree search wins while the pose still moves (stale seeds, long reach) and the grid
    // search once it has settled: both are enqueued every iteration and `tree_turn` says whose turn it is.  Set by the
    // host for the first search (no seeds: tree) and by the solve kernel afterwards, from the same quantity the grid
    // kernels double their budget on: last step's translation + rotation x object size against a fraction of a cell.
    double turn_limit, turn_scale;   // OA_TURN_FRAC (0.1) x cell edge and largest |coordinate| of the grid in use
    int32_t tree_turn;
    int32_t mx2_identity;     // base matrix_world is exactly the identity: mx2 @ v == v bit for bit for finite v (pair_eval)
    // multi-device mailbox exchange: sequence number of this loop's iteration 0 minus one.  It grows from loop to loop
    // (the host adds iters + 2 per loop), so a slot left over from an earlier loop can never be mistaken for a post of
    // this one and the mailboxes never have to be cleared (clearing them would need a cross-device ordering of its own)
    unsigned long long seq_base;
    // Jacobi warm start (rotation_from_covariance): the right singular vectors of the last iteration's covariance.
    // H = sum b a^T ~ R (sum a a^T), so H^T H -- what V diagonalises -- hardly changes from one iteration to the next:
    // started from the last V a solve takes one sweep of small rotations instead of three (8 rotations -> 3).
    double jac_v[9];
    int32_t jac_valid, pad4;
};

// Squared local search radius for the query p (rounded up to float).  Derivation: the pair test measures
// |mx2 a - mx2 b| in float32; for real local distance D the real world distance is >= sigma_min(mx2) D, and the
// float32 evaluation of the two products, the difference and the length is off by < 64u (|mx2|_inf (|a|+|b|) + |t|).
// init_loop_state() folds thresh, sigma_min, the target's extent and those error terms into cut_a, cut_b (+1e-5).
__host__ __device__ inline float search_cutoff2(const DevState *st, float px, float py, float pz)
{
    if (!(st->cut_a < 1e300)) return INFINITY;
    const double pabs = fabs((double)px) + fabs((double)py) + fabs((double)pz);
    const double c = st->cut_a + st->cut_b * pabs;
    const double c2 = c * c * (1.0 + 1e-6);
    return c2 < 3.0e38 ? (float)c2 : INFINITY;
}

// ------------------------------------------------------------------------------------------------
// float32 "mathutils" arithmetic (host + device, identical bits)
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline void m4_mul_v3(const float *M, float x, float y, float z, float &ox, float &oy, float &oz)
{
    // per row: double accumulation of float products, w = 1  (mathutils column_vector_multiplication)
    float r[3];
    for (int row = 0; row < 3; ++row) {
        const float *m = M + 4 * row;
        float p0 = m[0] * x, p1 = m[1] * y, p2 = m[2] * z, p3 = m[3] * 1.0f;
        double acc = 0.0;
        acc += (double)p0; acc += (double)p1; acc += (double)p2; acc += (double)p3;
        r[row] = (float)acc;
    }
    ox = r[0]; oy = r[1]; oz = r[2];
}

// co_find = imx2 @ (mx1 @ p)  (functions/general.py:287), the query of every search.  With the identity as the base object's
// matrix (DevState::mx2_identity; its inverse is then the identity too) the second product returns its argument bit for bit
// -- up to the sign of a zero, which no distance, cell or pair can see; a non-finite coordinate fails the searches' finiteness
// tests either way -- and is skipped: 39 fp32/fp64 instructions per query.
__host__ __device__ inline void co_find(const DevState *__restrict__ st, float x, float y, float z, float &px, float &py, float &pz)
{
    float wx, wy, wz;
    m4_mul_v3(st->mx1, x, y, z, wx, wy, wz);                       // mx1 @ p
    if (st->mx2_identity) { px = wx; py = wy; pz = wz; return; }
    m4_mul_v3(st->imx2, wx, wy, wz, px, py, pz);
}

__host__ __device__ inline void m4_mul_m4(const float *A, const float *B, float *out)
{
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) { float p = A[4 * i + k] * B[4 * k + j]; acc += (double)p; }
            r[4 * i + j] = (float)acc;
        }
    for (int i = 0; i < 16; ++i) out[i] = r[i];
}

__host__ __device__ inline double v3_length(float x, float y, float z)
{
    double acc = 0.0;
    float pz = z * z, py = y * y, px = x * x;
    acc += (double)pz; acc += (double)py; acc += (double)px;   // last component first (mathutils dot_vn_vn)
    return sqrt(acc);
}

// Matrix.inverted() as Blender 3.2 computes it (mathutils_Matrix.c: matrix_invert_internal -> determinant_m4, adjoint_m4_m4,
// element / det; blenlib math_matrix.c), ALL in float, left to right, no fused multiply-add (this translation unit is compiled
// with -ffp-contract=off): bit-identical to oracle/oa_oracle.c: oo_mat4_inverted, which cites the routines.  mathutils stores a
// matrix column-major; m[i][j] below is Blender's matrix[i][j] = element (row j, column i) of the row-major argument, so the
// operand order inside every cofactor is Blender's.  false if the float determinant is exactly 0 (Blender raises ValueError).
__host__ __device__ inline float bl_det_m2(float a, float b, float c, float d) { return a * d - b * c; }
__host__ __device__ inline float bl_det_m3(float a1, float a2, float a3, float b1, float b2, float b3, float c1, float c2, float c3)
{
    return (a1 * bl_det_m2(b2, b3, c2, c3) - b1 * bl_det_m2(a2, a3, c2, c3)) + c1 * bl_det_m2(a2, a3, b2, b3);
}
// adjoint_m4_m4 + determinant_m4: R[i * 4 + j] = Blender's R[i][j]
__host__ __device__ inline void m4_adjoint_det(const float *Af, float R[16], float &det)
{
    const float a1 = Af[0], b1 = Af[4], c1 = Af[8], d1 = Af[12];    // m[0][0..3]: column 0
    const float a2 = Af[1], b2 = Af[5], c2 = Af[9], d2 = Af[13];
    const float a3 = Af[2], b3 = Af[6], c3 = Af[10], d3 = Af[14];
    const float a4 = Af[3], b4 = Af[7], c4 = Af[11], d4 = Af[15];
    det = (((a1 * bl_det_m3(b2, b3, b4, c2, c3, c4, d2, d3, d4) - b1 * bl_det_m3(a2, a3, a4, c2, c3, c4, d2, d3, d4)) +
            c1 * bl_det_m3(a2, a3, a4, b2, b3, b4, d2, d3, d4)) - d1 * bl_det_m3(a2, a3, a4, b2, b3, b4, c2, c3, c4));
    R[0]  = bl_det_m3(b2, b3, b4, c2, c3, c4, d2, d3, d4);
    R[4]  = -bl_det_m3(a2, a3, a4, c2, c3, c4, d2, d3, d4);
    R[8]  = bl_det_m3(a2, a3, a4, b2, b3, b4, d2, d3, d4);
    R[12] = -bl_det_m3(a2, a3, a4, b2, b3, b4, c2, c3, c4);
    R[1]  = -bl_det_m3(b1, b3, b4, c1, c3, c4, d1, d3, d4);
    R[5]  = bl_det_m3(a1, a3, a4, c1, c3, c4, d1, d3, d4);
    R[9]  = -bl_det_m3(a1, a3, a4, b1, b3, b4, d1, d3, d4);
    R[13] = bl_det_m3(a1, a3, a4, b1, b3, b4, c1, c3, c4);
    R[2]  = bl_det_m3(b1, b2, b4, c1, c2, c4, d1, d2, d4);
    R[6]  = -bl_det_m3(a1, a2, a4, c1, c2, c4, d1, d2, d4);
    R[10] = bl_det_m3(a1, a2, a4, b1, b2, b4, d1, d2, d4);
    R[14] = -bl_det_m3(a1, a2, a4, b1, b2, b4, c1, c2, c4);
    R[3]  = -bl_det_m3(b1, b2, b3, c1, c2, c3, d1, d2, d3);
    R[7]  = bl_det_m3(a1, a2, a3, c1, c2, c3, d1, d2, d3);
    R[11] = -bl_det_m3(a1, a2, a3, b1, b2, b3, d1, d2, d3);
    R[15] = bl_det_m3(a1, a2, a3, b1, b2, b3, c1, c2, c3);
}
__host__ __device__ inline bool m4_inverted(const float *Af, float *out)
{
    float R[16], det;
    m4_adjoint_det(Af, R, det);
    if (det == 0.0f) return false;
    // out (row-major) [j * 4 + i] = R[i][j] / det
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[j * 4 + i] = R[i * 4 + j] / det;
    return true;
}

// ------------------------------------------------------------------------------------------------
// 3x3 rotation from the covariance H = sum b a^T (Kabsch; functions/general.py:181-187).
// One-sided Jacobi gives H V = U S; the reflection-corrected rotation U diag(1,1,det(UV^T)) V^T equals
// [u1 u2 u1xu2][v1 v2 v1xv2]^T, so only the two leading singular triplets are needed (rank >= 2).
// ------------------------------------------------------------------------------------------------
// One Jacobi rotation of columns (P, Q) of g and v; returns whether the pair needed one.  P, Q are template
// parameters so that every index is static: on the device the arrays then live in registers instead of scratch
// memory (the solve is a single fp64 lane; a scratch round trip per element dominated it).
template <int P, int Q>
__host__ __device__ inline bool jacobi_rotate(double g[3][3], double v[3][3])
{
    const double npp = g[0][P] * g[0][P] + g[1][P] * g[1][P] + g[2][P] * g[2][P];
    const double nqq = g[0][Q] * g[0][Q] + g[1][Q] * g[1][Q] + g[2][Q] * g[2][Q];
    const double dpq = g[0][P] * g[0][Q] + g[1][P] * g[1][Q] + g[2][P] * g[2][Q];
    if (dpq == 0.0 || dpq * dpq <= 1e-30 * (npp * nqq)) return false;     // |dpq| <= 1e-15 |g_p||g_q| (~4.5 eps)
    // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (nqq - npp) / (2 dpq), written with one division:
    // t = 2 dpq sign(w) / (|w| + sqrt(w^2 + 4 dpq^2)), w = nqq - npp
    const double w = nqq - npp, d2 = 2.0 * dpq;
    double c, s;
#if defined(__HIP_DEVICE_COMPILE__)
    // On the device a rotation is a chain of dependent fp64 operations on ONE lane, and the IEEE sqrt and division
    // sequences were four fifths of it (~0.5 us per rotation, 8 rotations per solve: tools/solve_microbench.hip).  The
    // ANGLE only steers the convergence -- any (c, s) with c^2 + s^2 = 1 is an exact orthogonal update of g and v -- so
    // t comes from the hardware's reciprocal-square-root / reciprocal seeds plus one Newton step each (relative error
    // ~1e-13: the off-diagonal term this rotation is meant to cancel survives at that level and the next sweep's test
    // sees it far below its threshold), and c = (1 + t^2)^(-1/2) from the seed plus two Newton steps (full precision,
    // so that c^2 + s^2 = 1 to the last bits).  Out-of-range magnitudes take the IEEE formulas below.
    const double h2 = __builtin_fma(w, w, d2 * d2);
    if (h2 > 1e-280 && h2 < 1e280) {
        const double y = __builtin_amdgcn_rsq(h2);
        double hyp = h2 * y;
        hyp = __builtin_fma(0.5 * y, __builtin_fma(-hyp, hyp, h2), hyp);
        const double den = fabs(w) + hyp;
        double r = __builtin_amdgcn_rcp(den);
        r = __builtin_fma(r, __builtin_fma(-den, r, 1.0), r);
        const double t = (w >= 0.0 ? d2 : -d2) * r;
        const double q = __builtin_fma(t, t, 1.0);
        c = __builtin_amdgcn_rsq(q);
        c = c * __builtin_fma(-0.5 * q * c, c, 1.5);
        c = c * __builtin_fma(-0.5 * q * c, c, 1.5);
        s = c * t;
    } else
#endif
    {
        const double t = (w >= 0.0 ? d2 : -d2) / (fabs(w) + sqrt(w * w + d2 * d2));
        c = 1.0 / sqrt(1.0 + t * t);
        s = c * t;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 3; ++r) {
        const double gp = g[r][P], gq = g[r][Q], vp = v[r][P], vq = v[r][Q];
        g[r][P] = c * gp - s * gq; g[r][Q] = s * gp + c * gq;
        v[r][P] = c * vp - s * vq; v[r][Q] = s * vp + c * vq;
    }
    return true;
}

// column j of a 3x3 held in registers (select chain instead of a dynamic index)
__host__ __device__ inline double col3(const double m[3][3], int r, int j)
{
    return j == 0 ? m[r][0] : (j == 1 ? m[r][1] : m[r][2]);
}

// v_io (optional, 9 doubles, row-major): in = an orthogonal matrix to start from (the V of a similar covariance: then
// g = H V has nearly orthogonal columns already), out = the V this solve ends with.  Any orthogonal start gives the same
// singular vectors up to rounding.
__host__ __device__ inline void rotation_from_covariance(const double H[9], double R[9], double *v_io = nullptr, bool v_valid = false)
{
    double g[3][3], v[3][3];
    if (v_io && v_valid) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v[i][j] = v_io[3 * i + j];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) g[i][j] = (H[3 * i] * v[0][j] + H[3 * i + 1] * v[1][j]) + H[3 * i + 2] * v[2][j];
    } else {
        g[0][0] = H[0]; g[0][1] = H[1]; g[0][2] = H[2];
        g[1][0] = H[3]; g[1][1] = H[4]; g[1][2] = H[5];
        g[2][0] = H[6]; g[2][1] = H[7]; g[2][2] = H[8];
        v[0][0] = 1.0; v[0][1] = 0.0; v[0][2] = 0.0;
        v[1][0] = 0.0; v[1][1] = 1.0; v[1][2] = 0.0;
        v[2][0] = 0.0; v[2][1] = 0.0; v[2][2] = 1.0;
    }
    for (int sweep = 0; sweep < 64; ++sweep) {
        bool any = jacobi_rotate<0, 1>(g, v);
        any = jacobi_rotate<0, 2>(g, v) || any;
        any = jacobi_rotate<1, 2>(g, v) || any;
        if (!any) break;
    }
    if (v_io)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) v_io[3 * i + j] = v[i][j];
    const double sg0 = sqrt(g[0][0] * g[0][0] + g[1][0] * g[1][0] + g[2][0] * g[2][0]);
    const double sg1 = sqrt(g[0][1] * g[0][1] + g[1][1] * g[1][1] + g[2][1] * g[2][1]);
    const double sg2 = sqrt(g[0][2] * g[0][2] + g[1][2] * g[1][2] + g[2][2] * g[2][2]);
    int j1 = 0;
    if (sg1 > sg0) j1 = 1;
    if (sg2 > (j1 == 0 ? sg0 : sg1)) j1 = 2;
    int j2 = (j1 == 0) ? 1 : 0;
    for (int j = 0; j < 3; ++j) {
        const double sj = j == 0 ? sg0 : (j == 1 ? sg1 : sg2), s2 = j2 == 0 ? sg0 : (j2 == 1 ? sg1 : sg2);
        if (j != j1 && sj > s2) j2 = j;
    }
    const double s_j1 = j1 == 0 ? sg0 : (j1 == 1 ? sg1 : sg2), s_j2 = j2 == 0 ? sg0 : (j2 == 1 ? sg1 : sg2);
    double u1[3], u2[3], v1[3], v2[3];
    const double i_j1 = (s_j1 > 0.0) ? 1.0 / s_j1 : 0.0, i_j2 = (s_j2 > 0.0) ? 1.0 / s_j2 : 0.0;   // (side by side: independent)
    for (int r = 0; r < 3; ++r) {
        v1[r] = col3(v, r, j1); v2[r] = col3(v, r, j2);
        u1[r] = (s_j1 > 0.0) ? col3(g, r, j1) * i_j1 : (r == 0 ? 1.0 : 0.0);
        u2[r] = (s_j2 > 0.0) ? col3(g, r, j2) * i_j2 : 0.0;
    }
    if (!(s_j2 > 0.0)) {   // rank <= 1: the rotation is not unique; complete u2 deterministically
        int m = 0;
        if (fabs(u1[1]) < fabs(u1[m])) m = 1;
        if (fabs(u1[2]) < fabs(m == 0 ? u1[0] : u1[1])) m = 2;
        const double e[3] = { m == 0 ? 1.0 : 0.0, m == 1 ? 1.0 : 0.0, m == 2 ? 1.0 : 0.0 };
        const double d = e[0] * u1[0] + e[1] * u1[1] + e[2] * u1[2];
        double nn = 0.0;
        for (int r = 0; r < 3; ++r) { u2[r] = e[r] - d * u1[r]; nn += u2[r] * u2[r]; }
        nn = sqrt(nn);
        for (int r = 0; r < 3; ++r) u2[r] /= nn;
    }
    const double u3[3] = { u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0] };
    const double v3[3] = { v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0] };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = u1[i] * v1[j] + u2[i] * v2[j] + u3[i] * v3[j];
}

// Solve from the accumulated sums.  Returns false when K < 3 (the reference's ValueError).
// Horn's closed form (functions/general.py:191-206, usesvd=False): the rotation is the unit quaternion that is the eigenvector of
// the largest eigenvalue of the symmetric 4 x 4 matrix N built from the covariance (the reference hands N's lower triangle to
// numpy.linalg.eigh); here a cyclic Jacobi diagonalisation in fp64.  H[3 i + j] = sum v1_i v0_j, as rotation_from_covariance takes it.
__host__ __device__ inline void rotation_from_covariance_horn(const double H[9], double R[9])
{
    const double xx = H[0], yy = H[4], zz = H[8], xy = H[3], yz = H[7], zx = H[2], xz = H[6], yx = H[1], zy = H[5];
    double N[4][4] = { { xx + yy + zz, yz - zy, zx - xz, xy - yx },
                       { yz - zy, xx - yy - zz, xy + yx, zx + xz },
                       { zx - xz, xy + yx, yy - xx - zz, yz + zy },
                       { xy - yx, zx + xz, yz + zy, zz - xx - yy } };
    double V[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < 4; ++p) { diag += fabs(N[p][p]); for (int q = p + 1; q < 4; ++q) off += fabs(N[p][q]); }
        if (!(off > 1e-300) || off <= 1e-18 * diag) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (N[p][q] == 0.0) continue;
                const double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 4; ++k) { const double a = N[k][p], b = N[k][q]; N[k][p] = c * a - sn * b; N[k][q] = sn * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = N[p][k], b = N[q][k]; N[p][k] = c * a - sn * b; N[q][k] = sn * a + c * b; }
                for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
            }
    }
    int m = 0;
    for (int k = 1; k < 4; ++k) if (N[k][k] > N[m][m]) m = k;
    double q[4] = { V[0][m], V[1][m], V[2][m], V[3][m] };
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (!(n2 > 0.0)) { for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0; return; }
    const double f = sqrt(2.0 / n2);                                 // quaternion_matrix (:54-62): q *= sqrt(2 / n), then the outer products
    for (int k = 0; k < 4; ++k) q[k] *= f;
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.0 - y * y - z * z; R[1] = x * y - z * w;       R[2] = x * z + y * w;
    R[3] = x * y + z * w;       R[4] = 1.0 - x * x - z * z; R[5] = y * z - x * w;
    R[6] = x * z - y * w;       R[7] = y * z + x * w;       R[8] = 1.0 - x * x - y * y;
}

__host__ __device__ inline bool solve_from_sums(const double *s, const double pivot[3], bool with_scale, double M[16],
                                                double *v_io = nullptr, bool v_valid = false, bool horn = false)
{
    const double K = s[S_K];
    if (!(K >= 3.0)) return false;
    double ca[3], cb[3];
    const double inv_K = 1.0 / K;
    for (int i = 0; i < 3; ++i) { ca[i] = s[S_A + i] * inv_K; cb[i] = s[S_B + i] * inv_K; }     // centroids (:160,:164)
    double H[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[3 * i + j] = s[S_H + 3 * i + j] - K * cb[i] * ca[j];  // dot(v1c, v0c.T) (:181)
    double R[9];
    if (horn) rotation_from_covariance_horn(H, R);                  // usesvd=False (:191-206)
    else rotation_from_covariance(H, R, v_io, v_valid);
    double sc = 1.0;
    if (with_scale) {                                                                    // :208-212
        const double n0 = s[S_AA] - K * (ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]);
        const double n1 = s[S_BB] - K * (cb[0] * cb[0] + cb[1] * cb[1] + cb[2] * cb[2]);
        sc = sqrt(n1 / n0);
    }
    // back to un-pivoted coordinates: c0 = ca + pivot, c1 = cb + pivot;  t = c1 - sR c0   (:215)
    for (int i = 0; i < 3; ++i) {
        double t = cb[i] + pivot[i];
        for (int j = 0; j < 3; ++j) {
            M[4 * i + j] = sc * R[3 * i + j];
            t -= sc * R[3 * i + j] * (ca[j] + pivot[j]);
        }
        M[4 * i + 3] = t;
    }
    M[12] = 0.0; M[13] = 0.0; M[14] = 0.0; M[15] = 1.0;
    return true;
}

__host__ __device__ inline double rotation_angle_3x3(const double M[16])
{
    // |angle| of the rotation part (columns normalised first, so uniform scale does not matter)
    double c[3];
    for (int j = 0; j < 3; ++j) c[j] = sqrt(M[j] * M[j] + M[4 + j] * M[4 + j] + M[8 + j] * M[8 + j]);
    double tr = 0.0;
    for (int j = 0; j < 3; ++j) tr += (c[j] > 0.0) ? M[4 * j + j] / c[j] : 1.0;
    double x = (tr - 1.0) * 0.5;
    x = x > 1.0 ? 1.0 : (x < -1.0 ? -1.0 : x);
    return acos(x);
}

// ------------------------------------------------------------------------------------------------
// closest point on a triangle (surface mode, oa_tri.hpp): Blender's closest_on_tri_to_point_v3 restated in float32
// with explicit operation order and no fma; bit-identical to oracle/oa_oracle.c: oo_closest_on_tri
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline float dot3f(const float *a, const float *b)
{
    float s = a[0] * b[0];
    s = s + a[1] * b[1];
    s = s + a[2] * b[2];
    return s;
}

// Blender math_geom.c closest_on_tri_to_point_v3, float32: the oracle's operations on the oracle's operands, but WITHOUT its
// early returns.  A wavefront takes every branch some lane takes, and with seven ways out (three vertices, three edges, the
// inside) the branchy form cost 370 instructions per call, the tail behind each return included.  Here everything the regions
// need is computed once (six dot products, the three cross terms), the region is the first of the oracle's tests that holds,
// ONE division serves whichever region it is (its numerator and denominator are selected first: the same IEEE quotient of
// the same two floats), and the point is put together from selected operands: ~150 instructions, the same bits.
// (A region's own expressions are evaluated for lanes outside it too; what they produce there -- a 0 / 0, say -- is dropped.)
__host__ __device__ inline void closest_on_tri(const float *p, const float *a, const float *b, const float *c, float *r)
{
    // (scalars, not arrays: selects between array elements keep the arrays in scratch memory)
    const float ax = a[0], ay = a[1], az = a[2], bx = b[0], by = b[1], bz = b[2], cx = c[0], cy = c[1], cz = c[2];
    const float px = p[0], py = p[1], pz = p[2];
    const float abx = bx - ax, aby = by - ay, abz = bz - az, acx = cx - ax, acy = cy - ay, acz = cz - az;
    const float apx = px - ax, apy = py - ay, apz = pz - az, bpx = px - bx, bpy = py - by, bpz = pz - bz;
    const float cpx = px - cx, cpy = py - cy, cpz = pz - cz, cbx = cx - bx, cby = cy - by, cbz = cz - bz;
#define OA_DOT3(ux, uy, uz, vx, vy, vz) (((ux) * (vx) + (uy) * (vy)) + (uz) * (vz))          /* dot3f's order */
    const float d1 = OA_DOT3(abx, aby, abz, apx, apy, apz), d2 = OA_DOT3(acx, acy, acz, apx, apy, apz);
    const float d3 = OA_DOT3(abx, aby, abz, bpx, bpy, bpz), d4 = OA_DOT3(acx, acy, acz, bpx, bpy, bpz);
    const float d5 = OA_DOT3(abx, aby, abz, cpx, cpy, cpz), d6 = OA_DOT3(acx, acy, acz, cpx, cpy, cpz);
#undef OA_DOT3
    const float vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    const float d43 = d4 - d3, d56 = d5 - d6;
    // the oracle's tests, in the oracle's order: the first that holds names the region
    const bool at_a = d1 <= 0.0f && d2 <= 0.0f;
    const bool at_b = d3 >= 0.0f && d4 <= d3;
    const bool on_ab = vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f;
    const bool at_c = d6 >= 0.0f && d5 <= d6;
    const bool on_ac = vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f;
    const bool on_bc = va <= 0.0f && d43 >= 0.0f && d56 >= 0.0f;
    enum { AT_A, AT_B, ON_AB, AT_C, ON_AC, ON_BC, INSIDE };
    const int region = at_a ? AT_A : at_b ? AT_B : on_ab ? ON_AB : at_c ? AT_C : on_ac ? ON_AC : on_bc ? ON_BC : INSIDE;
    // the one quotient: d1 / (d1 - d3) on ab, d2 / (d2 - d6) on ac, (d4 - d3) / ((d4 - d3) + (d5 - d6)) on bc,
    // 1 / ((va + vb) + vc) inside
    const float num = region == ON_AB ? d1 : region == ON_AC ? d2 : region == ON_BC ? d43 : 1.0f;
    const float den = region == ON_AB ? d1 - d3 : region == ON_AC ? d2 - d6 : region == ON_BC ? d43 + d56 : (va + vb) + vc;
    const float q = num / den;
    const float s1 = region == INSIDE ? vb * q : q;                // inside: v = vb * denom
    const float s2 = vc * q;                                       //         w = vc * denom
    const bool from_b = region == ON_BC || region == AT_B, from_c = region == AT_C;
    const bool vertex = region == AT_A || region == AT_B || region == AT_C, inside = region == INSIDE;
    const bool along_ac = region == ON_AC, along_cb = region == ON_BC;
    // t: a + ab v | a + ac w | (c - b) w + b | a + ab v;   inside: t + ac w
#define OA_TRI_POINT(k, av, bv, cv, abv, acv, cbv)                                                       \
    {                                                                                                    \
        const float base = from_b ? (bv) : (from_c ? (cv) : (av));                                       \
        const float e1 = along_ac ? (acv) : (along_cb ? (cbv) : (abv));                                  \
        const float t = base + e1 * s1;                                                                  \
        const float in = t + (acv) * s2;                                                                 \
        r[k] = vertex ? base : (inside ? in : t);                                                        \
    }
    OA_TRI_POINT(0, ax, bx, cx, abx, acx, cbx)
    OA_TRI_POINT(1, ay, by, cy, aby, acy, cby)
    OA_TRI_POINT(2, az, bz, cz, abz, acz, cbz)
#undef OA_TRI_POINT
}

__host__ __device__ inline float tri_dist2(const float *p, const float *r)
{
    const float d[3] = { r[0] - p[0], r[1] - p[1], r[2] - p[2] };
    return dot3f(d, d);
}


#if defined(__HIPCC__)

// triangle image: 3 x float4 per triangle {ax,ay,az,bx} {by,bz,cx,cy} {cz,-,-,-}
__device__ __forceinline__ void load_tri(const float4 *__restrict__ tri9, long long t, float *a, float *b, float *c)
{
    const float4 u = tri9[3 * t], v = tri9[3 * t + 1], w = tri9[3 * t + 2];
    a[0] = u.x; a[1] = u.y; a[2] = u.z; b[0] = u.w; b[1] = v.x; b[2] = v.y; c[0] = v.z; c[1] = v.w; c[2] = w.x;
}

// ------------------------------------------------------------------------------------------------
// packing kernels (one-time)
// ------------------------------------------------------------------------------------------------
// source: gather xyz[vlist[(begin + i) * stride]] -> float4; the tail up to ns_pad repeats the last point
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_pack_source(const float *__restrict__ xyz, const long long *__restrict__ vlist, long long stride,
                              long long begin, const int *__restrict__ members, int ns, int ns_pad,
                              float4 *__restrict__ src4, int *__restrict__ sel_vertex)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns_pad) return;
    const int k = i < ns ? i : (ns > 0 ? ns - 1 : 0);
    const long long s = members ? (long long)members[k] : begin + (long long)k;   // position in the selection
    const long long v = vlist ? vlist[s * stride] : s * stride;
    float4 p;
    p.x = xyz[3 * v]; p.y = xyz[3 * v + 1]; p.z = xyz[3 * v + 2]; p.w = 0.f;
    src4[i] = p;
    sel_vertex[i] = (int)v;                       // which vertex this slot holds (normals are gathered with it)
}

__global__ void k_gather_rows3(const float *__restrict__ rows, const int *__restrict__ sel, int n, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long v = sel[i];
    out[3ll * i] = rows[3 * v]; out[3ll * i + 1] = rows[3 * v + 1]; out[3ll * i + 2] = rows[3 * v + 2];
}
#endif  // !OA_FAMILY_TU

// 30-bit Morton key of a source point inside the source bounding box (spatial sort of the source slots: lanes of a
// wave then work on neighbouring points, which turns the grid search's scattered reads into mostly shared lines)
__device__ __forceinline__ unsigned spread10(unsigned v)
{
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_morton_keys(const float4 *__restrict__ src4, int ns, float lx, float ly, float lz, float sx, float sy,
                              float sz, unsigned *__restrict__ keys, int *__restrict__ slots)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const float4 p = src4[i];
    const float fx = fminf(fmaxf((p.x - lx) * sx, 0.f), 1023.f), fy = fminf(fmaxf((p.y - ly) * sy, 0.f), 1023.f),
                fz = fminf(fmaxf((p.z - lz) * sz, 0.f), 1023.f);      // NaN -> 0 through fmaxf
    keys[i] = spread10((unsigned)fx) | (spread10((unsigned)fy) << 1) | (spread10((unsigned)fz) << 2);
    slots[i] = i;
}

// sorted slot i takes original slot perm[i]; the padding repeats the last sorted point
__global__ void k_apply_perm(const float4 *__restrict__ src4o, const int *__restrict__ selo, const int *__restrict__ perm,
                             int ns, int ns_pad, float4 *__restrict__ src4, int *__restrict__ sel)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns_pad) return;
    const int o = perm[i < ns ? i : (ns > 0 ? ns - 1 : 0)];
    src4[i] = src4o[o];
    sel[i] = selo[o];
}

__global__ void k_iota(int *__restrict__ a, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = i;
}

// A shard adopted from the whole-selection order (oa_icp.hip: adopt_shard): src4 / sel hold its n points in slot
// (Morton) order; inv[m] = the slot of the shard's m-th point in the caller's order.  Fills the caller-order copy and the
// slot -> caller-order permutation, and pads both images to ns_pad by repeating their last point -- the layout
// k_pack_source + k_apply_perm leave behind.
__global__ void k_finish_shard(const int *__restrict__ inv, int n, int ns_pad, float4 *__restrict__ src4, int *__restrict__ sel,
                               float4 *__restrict__ src4o, int *__restrict__ perm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns_pad) return;
    const int m = i < n ? i : n - 1;
    const int s = inv[m];
    src4o[i] = src4[s];
    if (i < n) perm[s] = i;
    else { src4[i] = src4[n - 1]; sel[i] = sel[n - 1]; }
}

// target: groups of 4 vertices as [x0..x3][y0..y3][z0..z3]; vertices past nt are +INF (never selected)
__global__ void k_pack_target(const float *__restrict__ xyz, int nt, int n_groups_pad, float4 *__restrict__ tg)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups_pad) return;
    float c[3][4];
    for (int k = 0; k < 4; ++k) {
        const long long v = 4ll * g + k;
        for (int a = 0; a < 3; ++a) c[a][k] = (v < nt) ? xyz[3 * v + a] : INFINITY;
    }
    for (int a = 0; a < 3; ++a) tg[3 * (long long)g + a] = make_float4(c[a][0], c[a][1], c[a][2], c[a][3]);
}

__global__ void k_fill_keys(unsigned long long *keys, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = KEY_EMPTY;
}

// everything a shard's slots start from, in ONE launch (it was six operations -- four memsets and two fills -- and an upload
// is bound by the host's ~5 us per enqueued operation): no seed (prev = -1, win = all ones: index -1, wsafe = all ones),
// empty keys, sel = 0 and the hand-over counters at zero.  Null pointers are skipped (oa_reset_seeds, a new target: the
// seeds only).
__global__ void k_init_slots(int *__restrict__ prev, float4 *__restrict__ win, uint2 *__restrict__ wsafe,
                             unsigned long long *__restrict__ keys, int *__restrict__ sel, int *__restrict__ todo_count, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && todo_count) { todo_count[0] = 0; todo_count[1] = 0; }
    if (i >= n) return;
    const float ones = __uint_as_float(0xFFFFFFFFu);
    if (prev) prev[i] = -1;
    if (win) win[i] = make_float4(ones, ones, ones, ones);
    if (wsafe) wsafe[i] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    if (keys) keys[i] = KEY_EMPTY;
    if (sel) sel[i] = 0;
}

__global__ void k_fill_int(int *a, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

// (d2, idx) keys -> separate arrays (first n_out entries); resets all n keys for the next search
__global__ void k_decode_keys(unsigned long long *__restrict__ keys, int n, int n_out, const int *__restrict__ perm,
                              long long *__restrict__ idx, float *__restrict__ d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = keys[i];
    keys[i] = KEY_EMPTY;
    if (i >= n_out) return;
    const int o = perm ? perm[i] : i;                 // outputs are in the caller's (vlist) order
    const uint32_t j = (uint32_t)key;
    if (idx) idx[o] = (j == IDX_NONE) ? -1ll : (long long)j;
    if (d2) d2[o] = __uint_as_float((uint32_t)(key >> 32));
}
#endif  // !OA_FAMILY_TU

// ------------------------------------------------------------------------------------------------
// k_nn_search
// ------------------------------------------------------------------------------------------------
// grid = (n_splits, ns_pad / (256*R)): the split index is the FAST grid dimension, so with n_splits a multiple of 8
// the dispatcher's block -> XCD round-robin (block b on XCD b % 8) pins each target split to one XCD and that XCD's
// 4 MiB L2 keeps its 1/8 of the target image resident for the whole launch.  Each thread owns R source points (registers); the workgroup streams its
// split of the target through a double-buffered LDS tile.  All 64 lanes of a wave read the SAME LDS address
// (broadcast, conflict-free), so one ds_read_b128 feeds 4 targets x R points x 64 lanes = 256 R pair evaluations.
//
// Per group of 4 targets and per point: 4 x (3 sub, 1 mul, 2 fma) + 2 min3 + 1 compare = 27 VALU ops; the index
// is only resolved inside the (rare, exec-masked) `improved` branch.  Strict `<` against the running best and a
// lowest-k scan inside the group give "lowest index wins ties" exactly as the oracle's linear scan does.
__device__ __forceinline__ float d2_metric(float px, float py, float pz, float qx, float qy, float qz)
{
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float t = dx * dx;
    t = __builtin_fmaf(dy, dy, t);
    t = __builtin_fmaf(dz, dz, t);
    return t;
}

// XCD-aware workgroup order for the one-thread-per-query searches.  The dispatcher deals workgroups out round-robin -- workgroup
// b runs on XCD b % 8 -- so with the plain order every XCD's L2 sees every eighth workgroup of the whole (Morton-ordered)
// query range: neighbours in space, which read the same cells, sit on eight different L2s.  With this order XCD x works
// through ONE contiguous eighth of the query range: virtual index = start(x) + b / 8, start(x) = x q + min(x, r) for
// gridDim.x = 8 q + r (a bijection of 0 .. gridDim.x - 1).  Everything a workgroup does -- its queries, its row of
// partials -- goes by the virtual index, so results are unchanged.  Measured (A/B, two rounds each): 1M <-> 1M 60.3 -> 59.5 us per
// iteration, the surface search (1M points, 1.96M triangles) 0.630 -> 0.615 ms settled, nothing in its cold regime; shards of
// fewer than 1024 workgroups lose a little (100k: 29.8 -> 30.7 us) and keep the plain order.
__device__ __forceinline__ int xcd_block_index()
{
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    if (nb < 1024) return b;
    const int x = b & 7, q = nb >> 3, r = nb & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

// The same in chunks (round 6, k_tri_search_grid while the pose still moves): an XCD's share of the launch is every eighth CHUNK of
// `chunk` consecutive workgroups instead of one contiguous eighth of them all.  The queries are in Morton order, their cost follows
// the geometry, and a contiguous eighth is one region of the object: on the first search of a cold start XCD 0's share took 825 us
// of slot time where the mean was 642, and the launch ended with it (1004 us; per-workgroup stamps, profiles/r06i) -- seven XCDs
// idle for a third of the launch.  Chunks hand every XCD the same mix of regions; once the pose has settled the contiguous share
// is the faster one again (neighbours read the same cell lists and triangles: 0.341 against 0.355 ms per search), so the search
// picks by the test its candidate budget already makes.  The last, partial round of chunks goes by the plain order.
__device__ __forceinline__ int xcd_block_index_chunked(int chunk)
{
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    if (nb < 1024) return b;
    const int round = 8 * chunk, full = nb / round * round;         // workgroups in whole rounds of 8 chunks
    if (b >= full) return b;
    const int x = b & 7, k = b >> 3;                                // XCD, and this workgroup's number on it
    return (k / chunk) * round + x * chunk + k % chunk;
}

// "The pose still moves": first search of a loop, or last iteration's translation + rotation x object size above `threshold`
// (local units) -- the test behind the grid searches' budget_moving
__device__ __forceinline__ bool pose_moving(const DevState *__restrict__ st, double scale, double threshold)
{
    if (st->n == 0) return true;
    if (!st->use_target) return false;
    const int last = (st->n + 4) % 5;
    return (st->ring_t[last] + st->ring_r[last] * scale) * st->local_per_world > threshold;
}

// Target split of workgroup blockIdx.x: tiles [x T / n, (x + 1) T / n) of the T tiles (of `tile_groups` groups each), n =
// gridDim.x.  Split sizes differ by at most one tile, every split is non-empty (n <= T), and n stays what the host chose -- a
// multiple of 8 above 8, so that the dispatcher's round-robin (workgroup b on XCD b % 8) pins every split to ONE XCD, whose
// L2 then keeps its part of the image for the whole launch.  (Round 2 cut equal splits of ceil(T / n) tiles and launched
// ceil(T / that) of them: 70 instead of 72 for a 250k-point shard, 123 instead of 136 for 125k -- every XCD then saw every
// split, the 12 MB image no longer stayed in the L2s and the HBM-side traffic of a launch rose from 0.3 to 3.4 GB;
// profiles/r03d_bench_1Mx1M_pmc_summary.txt, the 4- and 8-shard passes.)
__device__ __forceinline__ void split_range(int n_groups_pad, int tile_groups, int &g_begin, int &g_end)
{
    const long long tiles = n_groups_pad / tile_groups;
    g_begin = (int)((long long)blockIdx.x * tiles / gridDim.x) * tile_groups;
    g_end = (int)((long long)(blockIdx.x + 1) * tiles / gridDim.x) * tile_groups;
}

__device__ __forceinline__ void split_range_of(int split, int n_splits, int n_groups_pad, int tile_groups, int &g_begin, int &g_end)
{
    const long long tiles = n_groups_pad / tile_groups;
    g_begin = (int)((long long)split * tiles / n_splits) * tile_groups;
    g_end = (int)((long long)(split + 1) * tiles / n_splits) * tile_groups;
}

template <int R>
__global__ __launch_bounds__(NN_THREADS) void k_nn_search(const DevState *__restrict__ st,
                                                          const float4 *__restrict__ src4,
                                                          const float4 *__restrict__ tg,
                                                          int n_groups_pad, unsigned long long *__restrict__ keys)
{
    if (st->halt) return;
    __shared__ float4 tile[2][TILE_GROUPS * 3];
    const int tid = threadIdx.x;
    const int base = blockIdx.y * (NN_THREADS * R);

    float px[R], py[R], pz[R], best[R];
    uint32_t bidx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 p = src4[base + r * NN_THREADS + tid];
        co_find(st, p.x, p.y, p.z, px[r], py[r], pz[r]);    // imx2 @ (...) = co_find (general.py:287)
        best[r] = INFINITY;
        bidx[r] = IDX_NONE;
    }

    int g_begin, g_end;
    split_range(n_groups_pad, TILE_GROUPS, g_begin, g_end);
    const int n_tiles = (g_end - g_begin) / TILE_GROUPS;           // splits are whole tiles by construction
    const float4 *tsrc = tg + 3ll * g_begin;

    // prologue: tile 0
    float4 s0 = tsrc[tid], s1 = tsrc[NN_THREADS + tid], s2 = tsrc[2 * NN_THREADS + tid];
    tile[0][tid] = s0; tile[0][NN_THREADS + tid] = s1; tile[0][2 * NN_THREADS + tid] = s2;
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < n_tiles);
        if (more) {                                               // next tile: global -> registers, hidden under compute
            const float4 *nsrc = tsrc + 3ll * TILE_GROUPS * (t + 1);
            s0 = nsrc[tid]; s1 = nsrc[NN_THREADS + tid]; s2 = nsrc[2 * NN_THREADS + tid];
        }
        const uint32_t jbase = (uint32_t)(g_begin + t * TILE_GROUPS) * 4u;
#pragma unroll 2
        for (int g = 0; g < TILE_GROUPS; ++g) {
            const float4 X = tile[cur][3 * g], Y = tile[cur][3 * g + 1], Z = tile[cur][3 * g + 2];
            const uint32_t j = jbase + 4u * (uint32_t)g;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d0 = d2_metric(px[r], py[r], pz[r], X.x, Y.x, Z.x);
                const float d1 = d2_metric(px[r], py[r], pz[r], X.y, Y.y, Z.y);
                const float d2 = d2_metric(px[r], py[r], pz[r], X.z, Y.z, Z.z);
                const float d3 = d2_metric(px[r], py[r], pz[r], X.w, Y.w, Z.w);
                const float m = __builtin_fminf(__builtin_fminf(d0, d1), d2);
                const float nb = __builtin_fminf(__builtin_fminf(m, d3), best[r]);
                if (nb < best[r]) {                                // rare after the first tiles
                    const uint32_t k = (d0 == nb) ? 0u : (d1 == nb) ? 1u : (d2 == nb) ? 2u : 3u;
                    bidx[r] = j + k;
                    best[r] = nb;
                }
            }
        }
        if (more) {
            tile[cur ^ 1][tid] = s0; tile[cur ^ 1][NN_THREADS + tid] = s1; tile[cur ^ 1][2 * NN_THREADS + tid] = s2;
        }
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
        unsigned long long *dst = keys + base + r * NN_THREADS + tid;
        if (gridDim.x == 1) *dst = key;
        else atomicMin(dst, key);                                  // (d2, idx) lexicographic: lowest index on ties
    }
}

// ------------------------------------------------------------------------------------------------
// k_nn_search_filtered : same answers as k_nn_search, ~3 VALU ops per pair instead of ~6.8
// ------------------------------------------------------------------------------------------------
// Idea: a cheap score that can only be used to PROVE "this target cannot win or tie", never to pick a winner.
//   centred coordinates   qh = fl32(q - c),  ph = fl32(p - c)          (c = centre of the target bbox)
//   score                 s_j = fma(ph.x, -2qh.x, fma(ph.y, -2qh.y, fma(ph.z, -2qh.z, w_j))),  w_j = fl32(|qh_j|^2)
//                         ~ |ph - qh_j|^2 - |ph|^2                      (3 fma per pair instead of 3 sub + mul + 2 fma)
//   threshold             T = round_up( best*(1+16u) + 16u*G^2 - |ph|^2 + 1e-30 ),  G = |ph| + max_j |qh_j|,  u = 2^-24
// Claim: s_j > T  implies  d2_metric(p, q_j) > best  (strictly), so a group of 4 targets whose smallest score exceeds
// T is skipped; every other group takes the exact path (difference-form d2, lexicographic (d2, index) update).
// Proof sketch (full derivation in docs/HISTORY.md 4.1): |s_j - S_j| <= 4.01u G^2 (three fma roundings + the rounding of w_j);
// centring moves |ph-qh_j| by at most 1.01u G; the exact metric is within a factor (1 +- 5.01u) of the real squared
// distance.  8u would already suffice for both terms; 16u is used.  T is rounded towards +inf.
// Targets/queries with non-finite or astronomically large coordinates disable the filter on the host side
// (k_nn_search is used instead).
//
// Seeding: `prev` holds each point's nearest index from the previous ICP iteration; its exact distance under the
// new transform is a valid upper bound for the minimum, so the scan starts with a tight threshold and the exact
// path is taken only for real improvements and near-ties.  Any seed gives the same final answer.
constexpr double FILTER_K = 16.0 * 5.9604644775390625e-08;   // 16 u
constexpr double FILTER_ABS = 1e-30;
constexpr int FTILE_GROUPS = 256;                            // groups per LDS tile: 256 x 64 B = 16 KiB

__device__ __forceinline__ float round_up_to_float(double x)
{
    float f = (float)x;
    if ((double)f < x) {                                      // nextafter(f, +inf)
        uint32_t b = __float_as_uint(f);
        if (f > 0.f) b += 1u;
        else if (f < 0.f) b -= 1u;
        else b = 1u;
        f = __uint_as_float(b);
    }
    return f;
}

// thr3 bounds the 3-D score, thr2 the 2-D score (the kept-plane distance can only be smaller than the 3-D one, the
// rounding analysis is the same with one fma less, and G bounds the 2-D norms as well)
__device__ __forceinline__ void filter_thresholds(float best, float hu, float hv, float hd, double qmax, float &thr2,
                                                  float &thr3)
{
    if (!(best < INFINITY)) { thr2 = INFINITY; thr3 = INFINITY; return; }   // nothing known yet: nothing can be skipped
    const double P2 = (double)hu * (double)hu + (double)hv * (double)hv;
    const double P3 = P2 + (double)hd * (double)hd;
    const double G = sqrt(P3) * (1.0 + 1e-12) + qmax;
    const double base = (double)best * (1.0 + FILTER_K) + FILTER_K * G * G + FILTER_ABS;
    thr2 = round_up_to_float(base - P2);
    thr3 = round_up_to_float(base - P3);
}

// filter images of the target, per group of 4 vertices.  (u, v, d) = fax: the axis of smallest extent is dropped by
// the first-level score.   tf2: [-2qu][-2qv][qu^2+qv^2]  (tiled through LDS)     tf3: [-2qd][|q|^2]  (rare path, global)
// Padding can never pass: its W2 / W3 are 3e38.
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_pack_filter(const float *__restrict__ xyz, int nt, int n_groups_pad, float cx, float cy, float cz,
                              int au, int av, int ad, float4 *__restrict__ tf2, float4 *__restrict__ tf3,
                              double *__restrict__ block_max_q2)
{
    __shared__ double red[4];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    double mx = 0.0;
    if (g < n_groups_pad) {
        float a[3][4], w2[4], w3[4];
        for (int k = 0; k < 4; ++k) {
            const long long v = 4ll * g + k;
            if (v < nt) {
                float q[3];
                q[0] = (float)((double)xyz[3 * v] - (double)cx);
                q[1] = (float)((double)xyz[3 * v + 1] - (double)cy);
                q[2] = (float)((double)xyz[3 * v + 2] - (double)cz);
                const double p2 = (double)q[au] * (double)q[au] + (double)q[av] * (double)q[av];
                const double q2 = p2 + (double)q[ad] * (double)q[ad];
                a[0][k] = -2.0f * q[au]; a[1][k] = -2.0f * q[av]; a[2][k] = -2.0f * q[ad];
                w2[k] = (float)p2;
                w3[k] = (float)q2;
                if (q2 > mx) mx = q2;
            } else {
                a[0][k] = 0.f; a[1][k] = 0.f; a[2][k] = 0.f; w2[k] = 3.0e38f; w3[k] = 3.0e38f;
            }
        }
        if (tf2) {                                                   // (nullptr: only the maximum is wanted -- the images are k_nn_search_filtered's, an OA_EXPERIMENTS kernel)
            tf2[3ll * g] = make_float4(a[0][0], a[0][1], a[0][2], a[0][3]);
            tf2[3ll * g + 1] = make_float4(a[1][0], a[1][1], a[1][2], a[1][3]);
            tf2[3ll * g + 2] = make_float4(w2[0], w2[1], w2[2], w2[3]);
            tf3[2ll * g] = make_float4(a[2][0], a[2][1], a[2][2], a[2][3]);
            tf3[2ll * g + 1] = make_float4(w3[0], w3[1], w3[2], w3[3]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(mx, off, 64); mx = o > mx ? o : mx; }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = red[0];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) m = red[k] > m ? red[k] : m;
        block_max_q2[blockIdx.x] = m;
    }
}

// per-block bounding box of the target (min xyz, max xyz); NaN-poisoned if any coordinate is not finite
__global__ void k_bbox_partial(const float *__restrict__ xyz, int nt, float *__restrict__ out /* blocks x 6 */)
{
    __shared__ float red[4][6];
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += gridDim.x * blockDim.x)
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3ll * i + a];
            if (!(fabsf(v) < INFINITY)) bad = true;
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
    if (bad) { lo[0] = NAN; hi[0] = NAN; }
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float l = __shfl_down(lo[a], off, 64), h = __shfl_down(hi[a], off, 64);
            lo[a] = (l != l || lo[a] != lo[a]) ? NAN : fminf(lo[a], l);
            hi[a] = (h != h || hi[a] != hi[a]) ? NAN : fmaxf(hi[a], h);
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = red[0][a];
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
            const float o = red[k][a];
            v = (o != o || v != v) ? NAN : (a < 3 ? fminf(v, o) : fmaxf(v, o));
        }
        out[blockIdx.x * 6 + a] = v;
    }
}
#endif  // !OA_FAMILY_TU

typedef float v2f __attribute__((ext_vector_type(2)));

template <int R, int TG = FTILE_GROUPS>
__global__ __launch_bounds__(NN_THREADS, (R <= 4 ? 4 : 2)) void k_nn_search_filtered(const DevState *__restrict__ st,
                                                                   const float4 *__restrict__ src4,
                                                                   const float4 *__restrict__ tg,
                                                                   const float4 *__restrict__ tf2,
                                                                   const float4 *__restrict__ tf3,
                                                                   const float4 *__restrict__ win,
                                                                   int n_groups_pad,
                                                                   unsigned long long *__restrict__ keys)
{
    // scalar v_fma_f32 throughout: the packed form (v_pk_fma_f32) measured 3-5 % slower in this loop (docs/HISTORY.md 5)
    if (st->halt) return;
    // TG = groups of 4 targets per LDS tile: 256 for large targets, 64 for small ones (more, shorter workgroups)
    constexpr int TILE_F4 = TG * 3, LOADS = (TILE_F4 + NN_THREADS - 1) / NN_THREADS;
    // whole tiles per thread (TG = 256): the staging registers must not be predicated, or the compiler parks them in
    // scratch and the tile loop writes and re-reads 48 B per lane per tile through HBM (PMC: 4.9 GB per launch)
    constexpr bool FULL = (TILE_F4 % NN_THREADS) == 0;
    __shared__ float4 tile[2][TILE_F4];
    const int tid = threadIdx.x;
    const double qmax = st->qmax;
    const float cx = st->tc[0], cy = st->tc[1], cz = st->tc[2];
    const int au = st->fax[0], av = st->fax[1];
    const int base = blockIdx.y * (NN_THREADS * R);
    float px[R], py[R], pz[R], hu[R], hv[R], hd[R], best[R], thr2[R], thr3[R], seed_d[R];
    uint32_t bidx[R], seed_idx[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = base + r * NN_THREADS + tid;
        const float4 p = src4[i];
        co_find(st, p.x, p.y, p.z, px[r], py[r], pz[r]);    // co_find (general.py:287)
        const float h0 = (float)((double)px[r] - (double)cx);
        const float h1 = (float)((double)py[r] - (double)cy);
        const float h2 = (float)((double)pz[r] - (double)cz);
        hu[r] = au == 0 ? h0 : (au == 1 ? h1 : h2);               // kept axes of the 2-D score
        hv[r] = av == 0 ? h0 : (av == 1 ? h1 : h2);
        hd[r] = (au + av == 1) ? h2 : ((au + av == 2) ? h1 : h0); // the dropped axis is the remaining one
        best[r] = INFINITY;
        bidx[r] = IDX_NONE;
        // seed: the slot's winner record of the last search (coordinates + index, kept current by k_pair_accumulate) --
        // one coalesced 16-byte read per split where the index alone cost a scattered gather from the target array
        const float4 sw = win ? win[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (__float_as_int(sw.w) >= 0) {
            const float d = d2_metric(px[r], py[r], pz[r], sw.x, sw.y, sw.z);
            if (d < INFINITY) { best[r] = d; bidx[r] = (uint32_t)__float_as_int(sw.w); }
        }
        seed_idx[r] = bidx[r];
        seed_d[r] = best[r];
        filter_thresholds(best[r], hu[r], hv[r], hd[r], qmax, thr2[r], thr3[r]);
    }

    int g_begin, g_end;
    split_range(n_groups_pad, TG, g_begin, g_end);
    const int n_tiles = (g_end - g_begin) / TG;
    const float4 *tsrc = tf2 + 3ll * g_begin;

    // staging registers for the next tile: named scalars (an indexed array ends up in scratch, see FULL above)
    static_assert(LOADS <= 3, "k_nn_search_filtered: tile of at most 3 float4 per thread");
    float4 stg0 = make_float4(0.f, 0.f, 0.f, 0.f), stg1 = stg0, stg2 = stg0;
#define OA_STG_EACH(OP)                                                                     \
    do {                                                                                    \
        if (LOADS > 0 && (FULL || 0 * NN_THREADS + tid < TILE_F4)) { OP(0, stg0); }         \
        if (LOADS > 1 && (FULL || 1 * NN_THREADS + tid < TILE_F4)) { OP(1, stg1); }         \
        if (LOADS > 2 && (FULL || 2 * NN_THREADS + tid < TILE_F4)) { OP(2, stg2); }         \
    } while (0)
#define OA_STG_FIRST(k, reg) reg = tsrc[(k) * NN_THREADS + tid]; tile[0][(k) * NN_THREADS + tid] = reg
    OA_STG_EACH(OA_STG_FIRST);
#undef OA_STG_FIRST
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < n_tiles);
        if (more) {                                               // next tile: global -> registers, hidden under compute
            const float4 *nsrc = tsrc + 3ll * TG * (t + 1);
#define OA_STG_LOAD(k, reg) reg = nsrc[(k) * NN_THREADS + tid]
            OA_STG_EACH(OA_STG_LOAD);
#undef OA_STG_LOAD
        }
        const int gbase = g_begin + t * TG;
        // GW groups (4*GW targets) per skip test.  Level 1 (hot): 2 fma + min per pair in the kept plane; level 2 (rare):
        // the dropped axis is added for the points that passed; level 3 (rarer): the exact metric.
        constexpr int GW = 2;
        for (int g = 0; g < TG; g += GW) {
            float4 AU[GW], AV[GW], W2[GW];
#pragma unroll
            for (int k = 0; k < GW; ++k) {
                AU[k] = tile[cur][3 * (g + k)]; AV[k] = tile[cur][3 * (g + k) + 1]; W2[k] = tile[cur][3 * (g + k) + 2];
            }
            float gm[R];
            bool hit = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float m = INFINITY;
#pragma unroll
                for (int k = 0; k < GW; ++k) {
                    const float c0 = __builtin_fmaf(hu[r], AU[k].x, __builtin_fmaf(hv[r], AV[k].x, W2[k].x));
                    const float c1 = __builtin_fmaf(hu[r], AU[k].y, __builtin_fmaf(hv[r], AV[k].y, W2[k].y));
                    const float c2 = __builtin_fmaf(hu[r], AU[k].z, __builtin_fmaf(hv[r], AV[k].z, W2[k].z));
                    const float c3 = __builtin_fmaf(hu[r], AU[k].w, __builtin_fmaf(hv[r], AV[k].w, W2[k].w));
                    m = __builtin_fminf(m, __builtin_fminf(__builtin_fminf(c0, c1), __builtin_fminf(c2, c3)));
                }
                gm[r] = m;
                hit = hit || !(m > thr2[r]);
            }
            if (hit) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!(gm[r] > thr2[r])) {                         // level 2: full 3-D score for these 4*GW targets
                        float m3 = INFINITY;
                        for (int k = 0; k < GW; ++k) {
                            const float4 AD = tf3[2ll * (gbase + g + k)], W3 = tf3[2ll * (gbase + g + k) + 1];
                            const float c0 = __builtin_fmaf(hu[r], AU[k].x, __builtin_fmaf(hv[r], AV[k].x, __builtin_fmaf(hd[r], AD.x, W3.x)));
                            const float c1 = __builtin_fmaf(hu[r], AU[k].y, __builtin_fmaf(hv[r], AV[k].y, __builtin_fmaf(hd[r], AD.y, W3.y)));
                            const float c2 = __builtin_fmaf(hu[r], AU[k].z, __builtin_fmaf(hv[r], AV[k].z, __builtin_fmaf(hd[r], AD.z, W3.z)));
                            const float c3 = __builtin_fmaf(hu[r], AU[k].w, __builtin_fmaf(hv[r], AV[k].w, __builtin_fmaf(hd[r], AD.w, W3.w)));
                            m3 = __builtin_fminf(m3, __builtin_fminf(__builtin_fminf(c0, c1), __builtin_fminf(c2, c3)));
                        }
                        if (!(m3 > thr3[r])) {                        // level 3: cannot be ruled out, exact metric
                            float b = best[r];
                            uint32_t bi = bidx[r];
                            for (int k = 0; k < GW; ++k) {
                                const float4 *eg = tg + 3ll * (gbase + g + k);
                                const float4 X = eg[0], Y = eg[1], Z = eg[2];
                                const uint32_t j = (uint32_t)(gbase + g + k) * 4u;
                                const float e0 = d2_metric(px[r], py[r], pz[r], X.x, Y.x, Z.x);
                                const float e1 = d2_metric(px[r], py[r], pz[r], X.y, Y.y, Z.y);
                                const float e2 = d2_metric(px[r], py[r], pz[r], X.z, Y.z, Z.z);
                                const float e3 = d2_metric(px[r], py[r], pz[r], X.w, Y.w, Z.w);
                                if (e0 < b || (e0 == b && j < bi)) { b = e0; bi = j; }
                                if (e1 < b || (e1 == b && j + 1u < bi)) { b = e1; bi = j + 1u; }
                                if (e2 < b || (e2 == b && j + 2u < bi)) { b = e2; bi = j + 2u; }
                                if (e3 < b || (e3 == b && j + 3u < bi)) { b = e3; bi = j + 3u; }
                            }
                            if (b < best[r]) filter_thresholds(b, hu[r], hv[r], hd[r], qmax, thr2[r], thr3[r]);
                            best[r] = b;
                            bidx[r] = (b < INFINITY) ? bi : IDX_NONE;     // overflowed distances (+inf) never win
                        }
                    }
                }
            }
        }
        if (more) {
#define OA_STG_STORE(k, reg) tile[cur ^ 1][(k) * NN_THREADS + tid] = reg
            OA_STG_EACH(OA_STG_STORE);
#undef OA_STG_STORE
        }
        __syncthreads();
    }
#undef OA_STG_EACH

    // A split reports when it has something to say: a better vertex than the seed, or the seed itself when it lies in
    // this split's range (exactly one split owns it, so the key is never left empty).  With 24 splits and a settled
    // pose that is one 64-bit atomic per point instead of 24 (PMC: 187 MB of HBM-side writes per launch for 8 MB of keys).
    const uint32_t own_lo = (uint32_t)g_begin * 4u, own_hi = (uint32_t)g_end * 4u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
        unsigned long long *dst = keys + base + r * NN_THREADS + tid;
        if (gridDim.x == 1) *dst = key;
        else {
            const bool seeded = seed_idx[r] != IDX_NONE;
            const bool improved = bidx[r] != seed_idx[r] || best[r] != seed_d[r];
            const bool owner = seeded && seed_idx[r] >= own_lo && seed_idx[r] < own_hi;
            if (!seeded || improved || owner) atomicMin(dst, key);  // (d2, idx) lexicographic: lowest index on ties
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_nn_search_sorted : k_nn_search_filtered with one more level in front (round 5) -- 1.84 VALU instructions per pair instead of 3.0
// ------------------------------------------------------------------------------------------------
// Still every (source, target) pair gets its own arithmetic; what changes is the ORDER of the target and a cheaper first test.
//   The target's images are laid out in the order of the coordinate u along the cloud's longest axis (a 30-bit quantised key,
//   stable sort; the order only matters for speed -- any permutation gives the same answers), so an LDS tile of 1024 vertices is
//   a thin slab in u; the source points a wave owns are neighbours in space (Morton order).
//   level 0 (hot loop)  the 1-D gap  a_j = |fl32(qu_j - hu)|:  ONE two-operand subtract + the min tree per pair (the |.| rides on
//                       the min's source modifier).  With base the right-hand side of filter_thresholds (best inflated by every
//                       rounding allowance of the 2-D / 3-D scores), a pair is a proven loser when its REAL gap t = |qu_j - hu|
//                       has t^2 > base (the distance along one axis never exceeds the 3-D distance; no rounding at all on this
//                       side of the inequality).  fl32(qu_j - hu) = (qu_j - hu)(1 + e), |e| <= 2^-24, so
//                       a_j > T1 = round_up(sqrt(base) (1 + 2^-22))  =>  t >= a_j / (1 + 2^-24) > sqrt(base).
//                       (Flushed denormals only ever make a_j smaller: fewer pairs skipped, never a wrong one.)
//                       For all slabs but the few around the wave's own u range every pair fails here.
//   levels 1, 2, 3      as k_nn_search_filtered (2-D score, 3-D score, exact metric), reached by ~6 % of the blocks at 1M <-> 1M.
// Indices: position j of the sorted images holds original vertex tidx[j]; the exact path compares and reports ORIGINAL indices
// (lowest index on ties, as everywhere).  A split owns a seed when seed index mod splits = split (no position lookup).
// Tiles are visited middle-out from the slab nearest to the workgroup's first point: an unseeded search (the first iteration of a
// loop) then finds a tight best at once instead of sweeping towards it; with seeds the order is immaterial.
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_sort_keys_axis(const float *__restrict__ xyz, int nt, int axis, double lo, double scale, unsigned *__restrict__ keys,
                                 int *__restrict__ ids)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nt) return;
    double t = ((double)xyz[3ll * i + axis] - lo) * scale;
    t = t < 0.0 ? 0.0 : (t > 1073741823.0 ? 1073741823.0 : t);
    keys[i] = (unsigned)t;
    ids[i] = i;
}

// images in sorted order, per group of 4 positions: tfs [qu][-2qv][qu^2+qv^2] (LDS tiles; -2qu is exact from qu), tf3s [-2qd][|q|^2],
// tgs [x][y][z] (exact), tidx original indices.  Same centring and roundings as k_pack_filter / k_pack_target.
__global__ void k_pack_sorted(const float *__restrict__ xyz, int nt, int n_groups_pad, const int *__restrict__ order, float cx, float cy,
                              float cz, int au, int av, int ad, float4 *__restrict__ tfs, float4 *__restrict__ tf3s,
                              float4 *__restrict__ tgs, int4 *__restrict__ tidx)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups_pad) return;
    float a[3][4], qu[4], w2[4], w3[4], e[3][4];
    int id[4];
    for (int k = 0; k < 4; ++k) {
        const long long j = 4ll * g + k;
        if (j < nt) {
            const int v = order[j];
            id[k] = v;
            float q[3];
            e[0][k] = xyz[3ll * v]; e[1][k] = xyz[3ll * v + 1]; e[2][k] = xyz[3ll * v + 2];
            q[0] = (float)((double)e[0][k] - (double)cx);
            q[1] = (float)((double)e[1][k] - (double)cy);
            q[2] = (float)((double)e[2][k] - (double)cz);
            const double p1 = (double)q[au] * (double)q[au];
            const double p2 = p1 + (double)q[av] * (double)q[av];
            const double q2 = p2 + (double)q[ad] * (double)q[ad];
            a[0][k] = -2.0f * q[au]; a[1][k] = -2.0f * q[av]; a[2][k] = -2.0f * q[ad];
            qu[k] = q[au]; w2[k] = (float)p2; w3[k] = (float)q2;
        } else {                                                  // padding can never pass a level, nor win
            id[k] = -1;
            e[0][k] = e[1][k] = e[2][k] = INFINITY;
            a[0][k] = a[1][k] = a[2][k] = 0.f;
            qu[k] = 1.0e18f;                                      // (far, and -2 qu stays finite; the tail of the order -- keys clamp)
            w2[k] = w3[k] = 3.0e38f;
        }
    }
    tfs[3ll * g] = make_float4(qu[0], qu[1], qu[2], qu[3]);
    tfs[3ll * g + 1] = make_float4(a[1][0], a[1][1], a[1][2], a[1][3]);
    tfs[3ll * g + 2] = make_float4(w2[0], w2[1], w2[2], w2[3]);
    tf3s[2ll * g] = make_float4(a[2][0], a[2][1], a[2][2], a[2][3]);
    tf3s[2ll * g + 1] = make_float4(w3[0], w3[1], w3[2], w3[3]);
    tgs[3ll * g] = make_float4(e[0][0], e[0][1], e[0][2], e[0][3]);
    tgs[3ll * g + 1] = make_float4(e[1][0], e[1][1], e[1][2], e[1][3]);
    tgs[3ll * g + 2] = make_float4(e[2][0], e[2][1], e[2][2], e[2][3]);
    tidx[g] = make_int4(id[0], id[1], id[2], id[3]);
}
#endif  // !OA_FAMILY_TU

// thr1: the 1-D gap's threshold (header above); thr2: the 2-D score's (filter_thresholds).  The 3-D score's is derived from thr2
// where it is needed.
__device__ __forceinline__ void sorted_thresholds(float best, float hu, float hv, float hd, double qmax, float &thr1, float &thr2)
{
    if (!(best < INFINITY)) { thr1 = INFINITY; thr2 = INFINITY; return; }
    const double P2 = (double)hu * (double)hu + (double)hv * (double)hv;
    const double P3 = P2 + (double)hd * (double)hd;
    const double G = sqrt(P3) * (1.0 + 1e-12) + qmax;
    const double base = (double)best * (1.0 + FILTER_K) + FILTER_K * G * G + FILTER_ABS;
    thr1 = round_up_to_float(sqrt(base) * (1.0 + 0x1p-22));
    thr2 = round_up_to_float(base - P2);
}

// Levels 1 .. 3 of the sorted kernels for ONE point and ONE group of 4 sorted positions (AU / AV / W2: the group's -2qu, -2qv,
// qu^2+qv^2).  Returns true when the point's best improved (its thresholds are then already renewed).
__device__ __forceinline__ bool sorted_finish_group(const float4 AU, const float4 AV, const float4 W2, long long group,
                                                    const float4 *__restrict__ tf3s, const float4 *__restrict__ tgs,
                                                    const int4 *__restrict__ tidx, float px, float py, float pz, float hu, float hv,
                                                    float hd, double qmax, float &best, uint32_t &bidx, float &thr1, float &thr2)
{
    // level 1: the 2-D score
    const float c0 = __builtin_fmaf(hu, AU.x, __builtin_fmaf(hv, AV.x, W2.x));
    const float c1 = __builtin_fmaf(hu, AU.y, __builtin_fmaf(hv, AV.y, W2.y));
    const float c2 = __builtin_fmaf(hu, AU.z, __builtin_fmaf(hv, AV.z, W2.z));
    const float c3 = __builtin_fmaf(hu, AU.w, __builtin_fmaf(hv, AV.w, W2.w));
    if (__builtin_fminf(__builtin_fminf(c0, c1), __builtin_fminf(c2, c3)) > thr2) return false;
    // level 2: the 3-D score
    const float4 AD = tf3s[2ll * group], W3 = tf3s[2ll * group + 1];
    const float e0s = __builtin_fmaf(hu, AU.x, __builtin_fmaf(hv, AV.x, __builtin_fmaf(hd, AD.x, W3.x)));
    const float e1s = __builtin_fmaf(hu, AU.y, __builtin_fmaf(hv, AV.y, __builtin_fmaf(hd, AD.y, W3.y)));
    const float e2s = __builtin_fmaf(hu, AU.z, __builtin_fmaf(hv, AV.z, __builtin_fmaf(hd, AD.z, W3.z)));
    const float e3s = __builtin_fmaf(hu, AU.w, __builtin_fmaf(hv, AV.w, __builtin_fmaf(hd, AD.w, W3.w)));
    // (the 3-D threshold from the 2-D one: base - P3 <= thr2 - hd^2, rounded up -- a register per point less than keeping it; a
    //  threshold that is too high only prunes less)
    if (__builtin_fminf(__builtin_fminf(e0s, e1s), __builtin_fminf(e2s, e3s)) > round_up_to_float((double)thr2 - (double)hd * (double)hd))
        return false;
    // level 3: cannot be ruled out, exact metric, original indices
    float b = best;
    uint32_t bi = bidx;
    const float4 *eg = tgs + 3ll * group;
    const float4 X = eg[0], Y = eg[1], Z = eg[2];
    const int4 J = tidx[group];
    const float e0 = d2_metric(px, py, pz, X.x, Y.x, Z.x);
    const float e1 = d2_metric(px, py, pz, X.y, Y.y, Z.y);
    const float e2 = d2_metric(px, py, pz, X.z, Y.z, Z.z);
    const float e3 = d2_metric(px, py, pz, X.w, Y.w, Z.w);
    if (e0 < b || (e0 == b && (uint32_t)J.x < bi)) { b = e0; bi = (uint32_t)J.x; }
    if (e1 < b || (e1 == b && (uint32_t)J.y < bi)) { b = e1; bi = (uint32_t)J.y; }
    if (e2 < b || (e2 == b && (uint32_t)J.z < bi)) { b = e2; bi = (uint32_t)J.z; }
    if (e3 < b || (e3 == b && (uint32_t)J.w < bi)) { b = e3; bi = (uint32_t)J.w; }
    const bool improved = b < best;
    if (improved) sorted_thresholds(b, hu, hv, hd, qmax, thr1, thr2);
    best = b;
    bidx = (b < INFINITY) ? bi : IDX_NONE;                        // overflowed distances (+inf) never win
    return improved;
}

// Which of a wave's 64 R slots each lane takes as its point r, for k_nn_search_sorted: the slots in the order of their u (the
// coordinate the target is sorted along) AT THE CURRENT POSE, so that point r of all lanes is the r-th slice of the wave's extent
// in u -- a quarter (R = 4) of the slabs a plain run of 64 consecutive slots reaches into (level 1 runs per point r and wave).
// One wave per 64 R slots; rank = number of slots with a smaller (key, position): a permutation whatever the values (NaN
// included).  order[first + rank] = position of the slot within the wave's range.
// (The same over a whole workgroup's 256 R slots -- sixteen thinner slices -- measured slower, 32.0 against 31.1 ms per iteration at
// 1M <-> 1M: the lanes of a point r then spread over the workgroup's whole extent in the other two axes and levels 2 / 3 run for more
// (wave, group) combinations.)
template <int R>
__global__ __launch_bounds__(64) void k_sorted_wave_order(const DevState *__restrict__ st, const float4 *__restrict__ src4, int au,
                                                          unsigned short *__restrict__ order)
{
    constexpr int WAVES = 1;
    constexpr int N = 64 * R * WAVES;
    __shared__ uint32_t key[N];
    const int t = threadIdx.x;
    const long long first = (long long)blockIdx.x * N;
    uint32_t mine[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 p = src4[first + r * (64 * WAVES) + t];
        float px, py, pz;
        co_find(st, p.x, p.y, p.z, px, py, pz);
        const float hu = au == 0 ? (float)((double)px - (double)st->tc[0]) : (au == 1 ? (float)((double)py - (double)st->tc[1]) : (float)((double)pz - (double)st->tc[2]));
        const uint32_t b = __float_as_uint(hu);
        mine[r] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);     // unsigned order = float order (NaNs at the ends: still a total order)
        key[r * (64 * WAVES) + t] = mine[r];
    }
    __syncthreads();
    int rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rank[r] = 0;
    for (int j = 0; j < N; ++j) {
        const uint32_t k = key[j];                                 // (broadcast read)
#pragma unroll
        for (int r = 0; r < R; ++r) rank[r] += (k < mine[r] || (k == mine[r] && j < r * (64 * WAVES) + t)) ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) order[first + rank[r]] = (unsigned short)(r * (64 * WAVES) + t);
}

// Seeds for the first search of a loop: every point against the ONE tile of the sorted images whose slab holds its own u (exact
// metric, original indices, lowest index on ties) -> keys.  1024 pairs per point instead of N_t; what it leaves is a distance really
// achieved (a valid starting best for k_nn_search_sorted pass 2), typically a few times the true nearest distance.
template <int TG>
__global__ __launch_bounds__(256) void k_nn_seed_sorted(const DevState *__restrict__ st, const float4 *__restrict__ src4, int ns_pad,
                                                        const float4 *__restrict__ tgs, const float4 *__restrict__ tfs,
                                                        const int4 *__restrict__ tidx, int n_groups_pad, int au,
                                                        unsigned long long *__restrict__ keys)
{
    if (st->halt) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns_pad) return;
    const float4 p = src4[i];
    float px, py, pz;
    co_find(st, p.x, p.y, p.z, px, py, pz);
    const float h[3] = { (float)((double)px - (double)st->tc[0]), (float)((double)py - (double)st->tc[1]), (float)((double)pz - (double)st->tc[2]) };
    const float hu = au == 0 ? h[0] : (au == 1 ? h[1] : h[2]);
    const int tiles = n_groups_pad / TG;
    int a = 0, b = tiles;                                          // first tile that starts beyond hu
    while (a < b) { const int mid = (a + b) >> 1; if (tfs[3ll * TG * mid].x <= hu) a = mid + 1; else b = mid; }
    const long long g0 = (long long)(a > 0 ? a - 1 : 0) * TG;
    float best = INFINITY;
    uint32_t bi = IDX_NONE;
    for (int g = 0; g < TG; ++g) {
        const float4 *eg = tgs + 3ll * (g0 + g);
        const float4 X = eg[0], Y = eg[1], Z = eg[2];
        const int4 J = tidx[g0 + g];
        const float e0 = d2_metric(px, py, pz, X.x, Y.x, Z.x), e1 = d2_metric(px, py, pz, X.y, Y.y, Z.y);
        const float e2 = d2_metric(px, py, pz, X.z, Y.z, Z.z), e3 = d2_metric(px, py, pz, X.w, Y.w, Z.w);
        if (e0 < best || (e0 == best && (uint32_t)J.x < bi)) { best = e0; bi = (uint32_t)J.x; }
        if (e1 < best || (e1 == best && (uint32_t)J.y < bi)) { best = e1; bi = (uint32_t)J.y; }
        if (e2 < best || (e2 == best && (uint32_t)J.z < bi)) { best = e2; bi = (uint32_t)J.z; }
        if (e3 < best || (e3 == best && (uint32_t)J.w < bi)) { best = e3; bi = (uint32_t)J.w; }
    }
    if (best < INFINITY && bi != IDX_NONE) {                       // (pads: +inf coordinates, index -1 -- never a seed)
        const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | bi;
        if (key < keys[i]) keys[i] = key;                          // (keys: KEY_EMPTY, or what an earlier pass left)
    }
}

// The work queue of k_nn_search_sorted (round 6).  One workgroup per (split, block of 1024 source points) in launch order left the chip
// 14 % (125k-point shard) to 9 % (1M points) short of the sum of its workgroups' own times (per-workgroup stamps, profiles/r06h):
// a workgroup whose points have their own slabs in its split runs levels 1-3 and lives 3.5x as long as one that only proves
// everything away at level 0 (a third of all slot time is theirs); with the hardware handing workgroups to the XCDs in strict
// rotation, slots stand empty behind a full XCD (6-8 % of them at any time), and the launch ends on a few long workgroups (the last
// twentieth of its span runs a third full).  So: as many workgroups as the chip holds, each taking (split, block) items off a
// queue until none is left -- block after block, a block's items in the order of the distance of the split from the block's own slab
// (its long items first).  NOT the long items of all blocks first: a long item waits on memory, a short one saturates the vector
// ALU, and a CU that holds both kinds does better than two CUs with one kind each (all long items first: 4.27 instead of 3.99 ms for
// the 125k-point shard, short items 58 us beside long ones and 75 us among themselves).  Eight queues, one per residue of the split
// index mod 8, and a workgroup serves the queue of its own XCD first (XCC_ID): a split's tiles stay in one L2, as they did when the
// launch order itself was the mapping.  The answers cannot depend on who does what: keys[] is merged with atomicMin.
// Measured (profiles/r06h): 1M <-> 1M 27.6 -> 26.6 ms per search, config 5's shard 58.7 -> 57.3, the 125k shard 3.98 -> 3.93.
// k_sorted_block_homes: per block of NN_THREADS x R slots the split that holds the u of its first point at the current pose, and
// the queues' counters back to zero (it runs in front of every queued launch).
constexpr int SORTED_QUEUES = 8, SORTED_QUEUE_STRIDE = 16;       // (counters 64 bytes apart)
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_sorted_block_homes(const DevState *__restrict__ st, const float4 *__restrict__ src4, int n_blocks, int block_slots,
                                     const float4 *__restrict__ tfs, int n_groups_pad, int tg, int au, int n_splits,
                                     int *__restrict__ homes, int *__restrict__ qcnt)
{
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y < SORTED_QUEUES) qcnt[y * SORTED_QUEUE_STRIDE] = 0;
    if (y >= n_blocks) return;
    const float4 p = src4[(long long)y * block_slots];
    float px, py, pz;
    co_find(st, p.x, p.y, p.z, px, py, pz);
    const float hu = au == 0 ? (float)((double)px - (double)st->tc[0]) : (au == 1 ? (float)((double)py - (double)st->tc[1]) : (float)((double)pz - (double)st->tc[2]));
    const int tiles = n_groups_pad / tg;
    int a = 0, b = tiles;                                          // first tile that starts beyond hu (NaN: tile 0 -- any home is a valid order)
    while (a < b) { const int mid = (a + b) >> 1; if (tfs[3ll * tg * mid].x <= hu) a = mid + 1; else b = mid; }
    const long long tile = a > 0 ? a - 1 : 0;
    int sp = (int)(tile * n_splits / tiles);                       // the split whose range [s tiles / S, (s + 1) tiles / S) holds the tile
    while (sp + 1 < n_splits && (long long)(sp + 1) * tiles / n_splits <= tile) ++sp;
    while (sp > 0 && (long long)sp * tiles / n_splits > tile) --sp;
    homes[y] = sp;
}
#endif  // !OA_FAMILY_TU

constexpr int SORT_ORDER_MAX = 1024;
#ifndef OA_SORTED_GW
#define OA_SORTED_GW 64                   // groups of 4 sorted vertices per skip test of k_nn_search_sorted (a build-time knob for sweeps)
#endif      // tiles of one split whose middle-out order fits the LDS table (more: ascending)

template <int R, int TG = FTILE_GROUPS>
__global__ __launch_bounds__(NN_THREADS, (R <= 4 ? 4 : 2)) void k_nn_search_sorted(const DevState *__restrict__ st,
                                                                 const float4 *__restrict__ src4,
                                                                 const float4 *__restrict__ tgs,
                                                                 const float4 *__restrict__ tfs,
                                                                 const float4 *__restrict__ tf3s,
                                                                 const int4 *__restrict__ tidx,
                                                                 const float4 *__restrict__ win,
                                                                 int n_groups_pad, int au, int av,
                                                                 unsigned long long *keys, int pass,
                                                                 const unsigned short *__restrict__ order,
                                                                 int n_splits, int n_blocks, const int *__restrict__ homes, int *qcnt)
{
    // pass 0: seeds from the winner records of the last accumulation (none on the first search of a loop: every split then has to
    // find a best of its own before it can skip anything).  pass 2: seeds from keys, where k_nn_seed_sorted has left every point's
    // nearest vertex within its own slab -- a distance really achieved, usually a few times the true one: enough for level 0.
    if (st->halt) return;
    constexpr int TILE_F4 = TG * 3, LOADS = (TILE_F4 + NN_THREADS - 1) / NN_THREADS;   // TG = 256: 3 whole rounds; TG = 64: 192 of 256 threads
    constexpr bool WHOLE = TILE_F4 % NN_THREADS == 0;
    static_assert(LOADS >= 1 && LOADS <= 3, "k_nn_search_sorted: 1 .. 3 float4 per thread and tile");
    __shared__ float4 tile[2][TILE_F4];
    __shared__ short ord[SORT_ORDER_MAX];
    __shared__ unsigned long long clk0[2];
    __shared__ int s_item[3];                                       // the workgroup's item {split, block}; [2]: the queues found empty so far
    const int tid0 = threadIdx.x;
    // qcnt == nullptr: one workgroup per (split, block), taken from the launch grid (n_splits = gridDim.x, n_blocks = gridDim.y)
    const bool queued = qcnt != nullptr;
    if (queued && tid0 == 0) s_item[2] = 0;
  for (;;) {
    // (the thread's index anew for every item: what is derived from it is then computed where it is used, as in the one-item form,
    //  instead of being hoisted out of this loop and held in registers -- or scratch -- through the scan)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    int split = (int)blockIdx.x, yblk = (int)blockIdx.y;
    if (queued) {
        if (tid == 0) {
            // SORTED_QUEUES queues when they divide the splits (else one): queue q holds the splits s = q + nq j; item p of a queue
            // is block p / per_q at rank p mod per_q, rank k being the split k steps (0, +1, -1, +2, ...) from the block's own
            const int nq = (n_splits % SORTED_QUEUES == 0) ? SORTED_QUEUES : 1, per_q = n_splits / nq;
            const int q_len = per_q * n_blocks;
            unsigned xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            int got = -1, got_y = 0;
            unsigned q_dry = (unsigned)s_item[2];
            for (int a = 0; a < nq && got < 0; ++a) {
                const int q = (int)((xcc + (unsigned)a) & (unsigned)(nq - 1));
                if (q_dry & (1u << q)) continue;
                const int pos = atomicAdd(qcnt + q * SORTED_QUEUE_STRIDE, 1);
                if (pos >= q_len) { q_dry |= 1u << q; continue; }
                got_y = pos / per_q;
                const int k = pos - got_y * per_q;
                int j0 = (homes[got_y] - q + nq / 2) / nq;          // the queue's split nearest to the block's own
                j0 = j0 < 0 ? 0 : (j0 >= per_q ? per_q - 1 : j0);
                int j = (j0 + ((k & 1) ? (k + 1) / 2 : -(k / 2))) % per_q;
                if (j < 0) j += per_q;
                got = q + nq * j;
            }
            s_item[0] = got; s_item[1] = got_y; s_item[2] = (int)q_dry;
        }
        __syncthreads();
        split = __builtin_amdgcn_readfirstlane(s_item[0]); yblk = __builtin_amdgcn_readfirstlane(s_item[1]);   // (scalar registers, as blockIdx was)
        if (split < 0) break;                                       // (the whole workgroup: nothing left anywhere)
    } else { n_splits = (int)gridDim.x; n_blocks = (int)gridDim.y; }
    const bool clk_wg = (split == 0 && yblk == n_blocks / 2 && tid == 0);   // one workgroup's item from the middle of the launch
    if (clk_wg) { clk0[0] = (unsigned long long)__builtin_readcyclecounter(); clk0[1] = wall_clock64(); }
    const double qmax = st->qmax;
    const float cx = st->tc[0], cy = st->tc[1], cz = st->tc[2];
    const int base = yblk * (NN_THREADS * R);
    // a wave owns R x 64 CONSECUTIVE slots (neighbours in space: the smallest extent in u, the fewest slabs it must look into);
    // point r of a lane is slot base + slot_of(r)
    // ... and WHICH of them is a lane's point r says k_sorted_wave_order: the wave's slots in the order of u, so that point r of all
    // lanes is one slice of the wave's extent (order == nullptr: slot r * 64 + lane)
    const int wave_first = base + (tid >> 6) * (64 * R);
#define OA_SLOT(r) (wave_first + (order ? (int)order[wave_first + (r) * 64 + (tid & 63)] : (r) * 64 + (tid & 63)))
    float px[R], py[R], pz[R], hu[R], hv[R], hd[R], best[R], thr1[R], thr2[R];
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
        const float4 sw = (win && pass == 0) ? win[i] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (__float_as_int(sw.w) >= 0) {
            const float d = d2_metric(px[r], py[r], pz[r], sw.x, sw.y, sw.z);
            if (d < INFINITY) { best[r] = d; bidx[r] = (uint32_t)__float_as_int(sw.w); }
        }
        if (pass == 2) {                                           // (a 64-bit load: whatever another split has merged by now is as good)
            const unsigned long long k = __hip_atomic_load(keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float kd = __uint_as_float((uint32_t)(k >> 32));
            if ((uint32_t)k != IDX_NONE && kd < INFINITY) { best[r] = kd; bidx[r] = (uint32_t)k; }
        }
        sorted_thresholds(best[r], hu[r], hv[r], hd[r], qmax, thr1[r], thr2[r]);
    }

    int g_begin, g_end;
    split_range_of(split, n_splits, n_groups_pad, TG, g_begin, g_end);
    const int n_tiles = (g_end - g_begin) / TG;
    const float4 *tsrc = tfs + 3ll * g_begin;

    // visiting order of this split's tiles: middle-out from the slab nearest (in u) to the workgroup's first point
    const bool ordered = n_tiles <= SORT_ORDER_MAX;
    if (ordered && tid == 0) {
        // the first vertex of tile t has u = tsrc[3 TG t].x, rising with t (the images are sorted by u): binary search for
        // the first tile that starts beyond the point -- the slab before it holds the point's u (an approximate start is fine)
        int a = 0, b = n_tiles;                                    // tiles [0, a) start at or before the point
        while (a < b) {
            const int mid = (a + b) >> 1;
            if (tsrc[3ll * TG * mid].x <= hu[0]) a = mid + 1; else b = mid;
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

    float4 stg0 = make_float4(0.f, 0.f, 0.f, 0.f), stg1 = stg0, stg2 = stg0;
#define OA_STG_ON(k) (WHOLE || (k) * NN_THREADS + tid < TILE_F4)
#define OA_STG_EACH(OP)                                       \
    do {                                                      \
        if (LOADS > 0 && OA_STG_ON(0)) { OP(0, stg0); }       \
        if (LOADS > 1 && OA_STG_ON(1)) { OP(1, stg1); }       \
        if (LOADS > 2 && OA_STG_ON(2)) { OP(2, stg2); }       \
    } while (0)
    {
        const float4 *fsrc = tsrc + 3ll * TG * OA_TILE_AT(0);
#define OA_STG_FIRST(k, reg) reg = fsrc[(k) * NN_THREADS + tid]; tile[0][(k) * NN_THREADS + tid] = reg
        OA_STG_EACH(OA_STG_FIRST);
#undef OA_STG_FIRST
    }
    __syncthreads();

    for (int t = 0; t < n_tiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < n_tiles);
        if (more) {                                               // next tile: global -> registers, hidden under compute
            const float4 *nsrc = tsrc + 3ll * TG * OA_TILE_AT(t + 1);
#define OA_STG_LOAD(k, reg) reg = nsrc[(k) * NN_THREADS + tid]
            OA_STG_EACH(OA_STG_LOAD);
#undef OA_STG_LOAD
        }
        const int gbase = g_begin + OA_TILE_AT(t) * TG;
        // 4 GW = 256 vertices x R points per skip test.  On gfx950 v_sub_f32 issues at full rate, v_min_f32, v_min3_f32 and v_cmp_*_f32
        // at half rate (tools/valu_rates.hip): the tree is all min3 (two comparisons per instruction), one compare per block.
        // Which blocks go on to level 1 is decided by the slabs' u, not by the block size (256 sorted vertices of a million span
        // 5e-4 of the axis, a point's reach 0.03), so larger blocks only save compares, branches and waits for the tile:
        // 16 / 32 / 64 / 128 / 256 / 512 vertices 31.1 / 28.8 / 28.0 / 27.6 / 27.3 / 27.4 ms per iteration at 1M <-> 1M.
        // The block is folded 16 vertices at a time, fully unrolled (the chain's value and the odd one out carried over; as a
        // loop of four chunks per trip: 28.3 ms): 16 tile registers live.
        constexpr int GW = OA_SORTED_GW < TG ? OA_SORTED_GW : TG;
        static_assert(GW % 4 == 0 && TG % GW == 0, "k_nn_search_sorted: blocks of whole chunks of 16 vertices");
        for (int g = 0; g < TG; g += GW) {
            float m[R], odd[R];
#pragma unroll
            for (int c = 0; c < GW / 4; ++c) {
                float4 QU[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) QU[k] = tile[cur][3 * (g + 4 * c + k)];
#pragma unroll
                for (int r = 0; r < R; ++r) {                      // level 0: one subtract + half a min3 per pair, no branch inside
                    float a[16];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a[4 * k] = __builtin_fabsf(QU[k].x - hu[r]); a[4 * k + 1] = __builtin_fabsf(QU[k].y - hu[r]);
                        a[4 * k + 2] = __builtin_fabsf(QU[k].z - hu[r]); a[4 * k + 3] = __builtin_fabsf(QU[k].w - hu[r]);
                    }
                    float v = c == 0 ? a[0] : __builtin_fminf(__builtin_fminf(m[r], odd[r]), a[0]);
#pragma unroll
                    for (int k = 1; k + 1 < 16; k += 2) v = __builtin_fminf(__builtin_fminf(v, a[k]), a[k + 1]);   // v_min3_f32
                    m[r] = v;
                    odd[r] = a[15];
                }
            }
            bool hit0 = false, hit[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                hit[r] = !(__builtin_fminf(m[r], odd[r]) > thr1[r]);
                hit0 |= hit[r];
            }
            if (!hit0) continue;
#pragma unroll 1
            for (int k = 0; k < GW; ++k) {                         // rare from here on: one group of 4 targets at a time
                const float4 Q = tile[cur][3 * (g + k)], AV = tile[cur][3 * (g + k) + 1], W2 = tile[cur][3 * (g + k) + 2];
                const float4 AU = make_float4(-2.0f * Q.x, -2.0f * Q.y, -2.0f * Q.z, -2.0f * Q.w);   // exact
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!hit[r]) continue;                         // (a point's run of 64 slots is a quarter of the wave's extent)
                    sorted_finish_group(AU, AV, W2, gbase + g + k, tf3s, tgs, tidx, px[r], py[r], pz[r], hu[r], hv[r], hd[r], qmax,
                                        best[r], bidx[r], thr1[r], thr2[r]);
                }
            }
        }
        if (more) {
#define OA_STG_STORE(k, reg) tile[cur ^ 1][(k) * NN_THREADS + tid] = reg
            OA_STG_EACH(OA_STG_STORE);
#undef OA_STG_STORE
        }
        __syncthreads();
    }
#undef OA_STG_EACH
#undef OA_STG_ON
#undef OA_TILE_AT

    if (clk_wg) {
        DevState *ws = const_cast<DevState *>(st);
        ws->search_clk[0] = (unsigned long long)__builtin_readcyclecounter() - clk0[0];
        ws->search_clk[1] = wall_clock64() - clk0[1];
    }
    // a split reports when it has something to say (k_nn_search_filtered); the seed's owner: seed index mod splits
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(best[r]) << 32) | bidx[r];
        unsigned long long *dst = keys + OA_SLOT(r);
        if (pass != 0) {                                           // merges into whatever keys holds: monotone
            if (key < __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(dst, key);
        } else if (n_splits == 1) *dst = key;
        else {
            // (the seed again, from the slot's record -- the same arithmetic as at the start -- instead of two registers per point
            //  held through the scan)
            uint32_t seed_idx = IDX_NONE;
            float seed_d = INFINITY;
            const float4 sw = win ? win[OA_SLOT(r)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (__float_as_int(sw.w) >= 0) {
                const float d = d2_metric(px[r], py[r], pz[r], sw.x, sw.y, sw.z);
                if (d < INFINITY) { seed_d = d; seed_idx = (uint32_t)__float_as_int(sw.w); }
            }
            const bool seeded = seed_idx != IDX_NONE;
            const bool improved = bidx[r] != seed_idx || best[r] != seed_d;
            const bool owner = seeded && (int)(seed_idx % (uint32_t)n_splits) == split;
            if (!seeded || improved || owner) atomicMin(dst, key);
        }
    }
    if (!queued) break;
  }
}

#undef OA_SLOT

// ------------------------------------------------------------------------------------------------
// k_pair_accumulate
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Optional normal-angle rejection (an EXTENSION: the reference has no such test -- SURVEY.md D3).  A pair is kept
// only if the angle between the world-space normals of the source vertex and of its correspondence is at most
// max_angle.  Normals are carried to world space with the inverse-transpose of the object matrices (imx1 / imx2).
struct NormalTest {
    const float *src_n;     // ns x 3, align-local, same slot order as src4 (nullptr = test disabled)
    const float *tgt_n;     // nt x 3 per target vertex (vertex mode); nullptr in surface mode = geometric face normal
    double cos_min;
};

__device__ __forceinline__ bool normal_angle_ok(const float *imx1, const float *imx2, const float *ns, const float *nt,
                                                double cos_min)
{
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) {      // (M^-1)^T n : column k of the inverse dotted with n
        a[k] = (double)imx1[k] * (double)ns[0] + (double)imx1[4 + k] * (double)ns[1] + (double)imx1[8 + k] * (double)ns[2];
        b[k] = (double)imx2[k] * (double)nt[0] + (double)imx2[4 + k] * (double)nt[1] + (double)imx2[8 + k] * (double)nt[2];
    }
    const double ab = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
    const double aa = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2];
    const double bb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
    const double c = ab / sqrt(aa * bb);
    return c >= cos_min;               // NaN (zero normal) -> rejected
}

// Wave-level sum of the 24 running sums: a reduce-scatter butterfly.  Each exchange step halves the number of values
// a lane still carries (24 -> 12 -> 6 -> 3, partner = lane ^ 32, ^ 16, ^ 8), then the remaining 3 values are summed
// over the 8 lanes that share bits 5..3 (^ 4, ^ 2, ^ 1): 30 double shuffles instead of 24 x 6 = 144.  Afterwards every
// lane holds, in out[0..2], the wave totals of sums number first .. first + 2.  Fixed pattern => bitwise reproducible.
__device__ __forceinline__ void wave_reduce_sums(const double (&acc)[NSUMS], int lane, double (&out)[3], int &first)
{
    double a12[12], a6[6];
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const double keep = b5 ? acc[12 + k] : acc[k], send = b5 ? acc[k] : acc[12 + k];
        a12[k] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double keep = b4 ? a12[6 + k] : a12[k], send = b4 ? a12[k] : a12[6 + k];
        a6[k] = keep + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double keep = b3 ? a6[3 + k] : a6[k], send = b3 ? a6[k] : a6[3 + k];
        double v = keep + __shfl_xor(send, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        out[k] = v;
    }
    first = (b5 ? 12 : 0) + (b4 ? 6 : 0) + (b3 ? 3 : 0);
}

struct PairOut {            // optional per-point outputs for the make_pairs contract
    unsigned char *valid;   // ns
    float  *b;              // ns x 3  (imx1 @ (mx2 @ co1))
    double *dist;           // ns
    int    *nn_idx;         // ns
    float  *nn_d2;          // ns
    const int *perm;        // sorted slot -> caller-order slot (nullptr = identity); outputs are in caller order
};

// ---- the pair test and its contribution to the sums: shared by k_pair_accumulate and by the search kernels that
// accumulate in their epilogue (same operations, same order, same bits) ------------------------------------------------
// (cx, cy, cz) = co_find, the source point in base-local space; (qx, qy, qz) = co1, its correspondence there;
// tn = the correspondence's normal (only read when the normal-angle extension is on); slot = the source slot (normals).
// Returns the reference's `dist < thresh` (functions/general.py:299-302) and, for a pair that passes, b = imx1 @ (mx2 @ co1).
__device__ __forceinline__ bool pair_eval(const DevState *__restrict__ st, float cx, float cy, float cz, float qx, float qy,
                                          float qz, const NormalTest &nrm, long long slot, const float *tn, double thresh,
                                          float &bx, float &by, float &bz, double &dist)
{
    float ax = cx, ay = cy, az = cz, wbx = qx, wby = qy, wbz = qz;
    if (!st->mx2_identity) {
        // (with the identity the products below return their argument bit for bit -- up to the sign of a zero, which neither
        //  the distance nor b can see -- and the two of them are a fifth of this function's fp64 work)
        m4_mul_v3(st->mx2, cx, cy, cz, ax, ay, az);     // mx2 @ co_find             (general.py:299)
        m4_mul_v3(st->mx2, qx, qy, qz, wbx, wby, wbz);  // mx2 @ co1                 (general.py:299)
    }
    dist = v3_length(ax - wbx, ay - wby, az - wbz);
    bool valid = dist < thresh;                        // face_index != -1 always holds (general.py:302)
    if (valid && nrm.src_n) {
        const float sn[3] = { nrm.src_n[3ll * slot], nrm.src_n[3ll * slot + 1], nrm.src_n[3ll * slot + 2] };
        valid = normal_angle_ok(st->imx1, st->imx2, sn, tn, nrm.cos_min);
    }
    bx = by = bz = 0.f;
    if (valid) m4_mul_v3(st->imx1, wbx, wby, wbz, bx, by, bz);   // imx1 @ (mx2 @ co1)  (general.py:304)
    return valid;
}

__device__ __forceinline__ void pair_add(double (&acc)[NSUMS], float p_x, float p_y, float p_z, float bx, float by, float bz,
                                         double dist, double pvx, double pvy, double pvz, double d_pivot)
{
    const double a0 = (double)p_x - pvx, a1 = (double)p_y - pvy, a2 = (double)p_z - pvz;
    const double b0 = (double)bx - pvx, b1 = (double)by - pvy, b2 = (double)bz - pvz;
    acc[S_A] += a0; acc[S_A + 1] += a1; acc[S_A + 2] += a2;
    acc[S_B] += b0; acc[S_B + 1] += b1; acc[S_B + 2] += b2;
    acc[S_H + 0] += b0 * a0; acc[S_H + 1] += b0 * a1; acc[S_H + 2] += b0 * a2;
    acc[S_H + 3] += b1 * a0; acc[S_H + 4] += b1 * a1; acc[S_H + 5] += b1 * a2;
    acc[S_H + 6] += b2 * a0; acc[S_H + 7] += b2 * a1; acc[S_H + 8] += b2 * a2;
    acc[S_AA] += (a0 * a0 + a1 * a1) + a2 * a2;
    acc[S_BB] += (b0 * b0 + b1 * b1) + b2 * b2;
    acc[S_K] += 1.0;
    const double dd = dist - d_pivot;
    acc[S_D] += dd;
    acc[S_DD] += dd * dd;
}

// every lane's acc -> the workgroup's totals -> one row of per-workgroup partials (wave butterfly, then the waves in
// order through LDS: a fixed pattern, bitwise reproducible).  red: __shared__ double[blockDim.x / 64][NSUMS].
// Called by ALL threads of the workgroup.
__device__ __forceinline__ void block_store_partial(const double (&acc)[NSUMS], double (*red)[NSUMS], double *__restrict__ row)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        double tot[3];
        int first;
        wave_reduce_sums(acc, lane, tot, first);
        if ((lane & 7) == 0) { red[wave][first] = tot[0]; red[wave][first + 1] = tot[1]; red[wave][first + 2] = tot[2]; }
    }
    __syncthreads();
    if (threadIdx.x < NSUMS) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
        row[threadIdx.x] = v;
    }
}

// The same for search kernels whose lanes hold ONE pair each (k_nn_search_grid<L, true>): the 20 sums are formed and
// reduced in two halves of 12 and 8, so that the epilogue of a kernel built for 6 waves per SIMD (80 VGPRs) never holds 24
// doubles plus the butterfly's temporaries at once.  Reduce-scatter over lane bits 5, 4 (, 3), then a plain butterfly.
// (a, b) = the pair relative to the pivot, dd = dist - d_pivot; lanes without a valid pair contribute zeros.
// The exchanges stay in the VALU (round 3; they were ds_bpermute pairs with a select on either side -- 62 LDS operations and
// as many selects per wave): gfx950's v_permlane32_swap / v_permlane16_swap ARE the halving step (the upper half of one
// register against the lower half of the other: each half keeps its value and receives the other half's copy of it), and the
// butterflies inside a row of 16 lanes are DPP moves.  Same additions between the same values as before (operands swapped in
// half of the lanes; IEEE addition commutes), so the rows keep their bits: tools/reduce_check.hip holds the old form.
__device__ __forceinline__ double f64_from(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

// lanes 0..31: a[l] + a[l + 32]     lanes 32..63: b[l - 32] + b[l]
__device__ __forceinline__ double halve_add32(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return f64_from(lo[0], hi[0]) + f64_from(lo[1], hi[1]);
}

// even rows of 16 lanes: a[l] + a[l + 16]     odd rows: b[l - 16] + b[l]
__device__ __forceinline__ double halve_add16(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return f64_from(lo[0], hi[0]) + f64_from(lo[1], hi[1]);
}

// t[l ^ X] inside a row of 16 lanes: X = 8 is a rotation by 8, X = 2 and 1 are quad permutations; X = 4 takes lane l + 4 in
// the banks {0..3, 8..11} and lane l - 4 in the others (two DPP moves into one register)
template <int X>
__device__ __forceinline__ double row_xor(double t)
{
    static_assert(X == 8 || X == 4 || X == 2 || X == 1, "");
    const int lo = __double2loint(t), hi = __double2hiint(t);
    int rl, rh;
    if (X == 4) {
        rl = __builtin_amdgcn_update_dpp(lo, lo, 0x104, 0xf, 0x5, false);      // row_shl:4 -> banks 0, 2
        rl = __builtin_amdgcn_update_dpp(rl, lo, 0x114, 0xf, 0xa, false);      // row_shr:4 -> banks 1, 3
        rh = __builtin_amdgcn_update_dpp(hi, hi, 0x104, 0xf, 0x5, false);
        rh = __builtin_amdgcn_update_dpp(rh, hi, 0x114, 0xf, 0xa, false);
    } else {
        constexpr int ctrl = X == 8 ? 0x128 : (X == 2 ? 0x4e : 0xb1);           // row_ror:8, quad_perm [2,3,0,1], [1,0,3,2]
        rl = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);
        rh = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);
    }
    return __hiloint2double(rh, rl);
}

template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], int lane, double *red_row)
{
    static_assert(N == 12 || N == 8, "halves of the 24 sums");
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
    double u[N / 2], w[3];                                          // (w: N / 4 values; 3 slots so that the N == 12 branch indexes in bounds when N == 8 is compiled)
#pragma unroll
    for (int k = 0; k < N / 2; ++k) u[k] = halve_add32(v[k], v[N / 2 + k]);     // lower half of the wave: v[k], upper: v[N/2 + k]
#pragma unroll
    for (int k = 0; k < N / 4; ++k) w[k] = halve_add16(u[k], u[N / 4 + k]);
    if (N == 12) {                                                   // 3 values left, 16 lanes share them
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double t = w[k];
            t += row_xor<8>(t); t += row_xor<4>(t); t += row_xor<2>(t); t += row_xor<1>(t);
            w[k] = t;
        }
        if ((lane & 15) == 0) {
            const int first = (b5 ? 6 : 0) + (b4 ? 3 : 0);
            red_row[first] = w[0]; red_row[first + 1] = w[1]; red_row[first + 2] = w[2];
        }
    } else {                                                         // 2 values left: one more halving, then 8 lanes share one
        const double keep = b3 ? w[1] : w[0], send = b3 ? w[0] : w[1];
        double t = keep + row_xor<8>(send);
        t += row_xor<4>(t); t += row_xor<2>(t); t += row_xor<1>(t);
        if ((lane & 7) == 0) red_row[(b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0)] = t;
    }
}

// red: __shared__ double[blockDim.x / 64][NSUMS].  Called by ALL threads of the workgroup.
// (Combining the workgroups' rows any further inside the launch -- groups of 16 meeting at a counter, rows and counter in
// agent-scope atomics -- was built and measured: the store -> acknowledge -> atomic -> load chain adds 4-5 us to the end
// of every launch, more than the wider reduction it saves; a reduction over several workgroups meeting at ONE counter
// costs +8 us with 16 of them and +16..23 us with 64, with or without fences.  So: one row per workgroup, one wide
// single-workgroup reduction behind it.)
__device__ __forceinline__ void block_store_pair(bool valid, double a0, double a1, double a2, double b0, double b1, double b2,
                                                 double dd, double (*red)[NSUMS], double *__restrict__ row,
                                                 long long *stamps = nullptr)      // instrumented builds: clock after the wave's part, after the barrier
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!valid) { a0 = a1 = a2 = b0 = b1 = b2 = dd = 0.0; }
    {
        double h[12] = { a0, a1, a2, b0, b1, b2, b0 * a0, b0 * a1, b0 * a2, b1 * a0, b1 * a1, b1 * a2 };   // sums 0 .. 11
        wave_reduce_scatter<12>(h, lane, &red[wave][0]);
    }
    {
        double h[8] = { b2 * a0, b2 * a1, b2 * a2, (a0 * a0 + a1 * a1) + a2 * a2, (b0 * b0 + b1 * b1) + b2 * b2,
                        valid ? 1.0 : 0.0, dd, dd * dd };                                                       // sums 12 .. 19
        wave_reduce_scatter<8>(h, lane, &red[wave][12]);
    }
    if (lane < 4) red[wave][20 + lane] = 0.0;                                                                   // reserved
    if (stamps) stamps[0] = (long long)__builtin_readcyclecounter();
    __syncthreads();
    if (stamps) stamps[1] = (long long)__builtin_readcyclecounter();
    if (threadIdx.x < NSUMS) {
        double v = red[0][threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v += red[w][threadIdx.x];
        row[threadIdx.x] = v;
    }
}

template <bool EMIT>
__global__ __launch_bounds__(ACC_THREADS) void k_pair_accumulate(const DevState *__restrict__ st,
                                                                 const float4 *__restrict__ src4, int ns,
                                                                 const float *__restrict__ tgt_xyz,
                                                                 unsigned long long *__restrict__ keys,
                                                                 int *__restrict__ prev, float4 *__restrict__ win,
                                                                 const float4 *__restrict__ tri9, NormalTest nrm,
                                                                 double *__restrict__ partials, PairOut out,
                                                                 unsigned long long *__restrict__ t_acc_start)
{
    __shared__ double red[ACC_THREADS / 64][NSUMS];
    if (t_acc_start && blockIdx.x == 0 && threadIdx.x == 0) *t_acc_start = wall_clock64();   // ~ the end of the search
    double acc[NSUMS];
#pragma unroll
    for (int k = 0; k < NSUMS; ++k) acc[k] = 0.0;
    const bool halted = st->halt != 0;
    const double thresh = st->thresh;
    const double pvx = st->pivot[0], pvy = st->pivot[1], pvz = st->pivot[2];
    const double d_pivot = st->d_pivot;

    if (!halted) {
        for (int i = blockIdx.x * ACC_THREADS + threadIdx.x; i < ns; i += gridDim.x * ACC_THREADS) {
            const unsigned long long key = keys[i];
            keys[i] = KEY_EMPTY;                                   // ready for the next iteration's atomicMin
            const uint32_t idx = (uint32_t)key;
            if (prev) prev[i] = (idx == IDX_NONE) ? -1 : (int)idx;  // seed for the next iteration's filtered search
            bool valid = false;
            float bx = 0.f, by = 0.f, bz = 0.f;
            double dist = 0.0;
            const float4 p = src4[i];
            // vertex mode: the slot's winner record (coordinates + index).  The grid and tree searches leave this
            // search's winner there; after a brute-force search it still holds the previous winner, which in a
            // converging loop is mostly the same vertex.  A matching index saves the gather from the target array.
            float4 wrec = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (win) wrec = win[i];
            if (idx != IDX_NONE) {
                float cx, cy, cz;
                co_find(st, p.x, p.y, p.z, cx, cy, cz);       // co_find                   (general.py:287)
                float qx, qy, qz;                                    // co1 (general.py:297)
                float tn[3] = { 0.f, 0.f, 0.f };                     // correspondence normal (only for the extension)
                if (tri9) {                                          // surface mode: closest point on triangle `idx`
                    float ta[3], tb[3], tc[3], rr[3];
                    const float cf[3] = { cx, cy, cz };
                    load_tri(tri9, idx, ta, tb, tc);
                    closest_on_tri(cf, ta, tb, tc, rr);
                    qx = rr[0]; qy = rr[1]; qz = rr[2];
                    if (nrm.src_n) {                                 // geometric face normal (Blender normal_tri_v3 order)
                        const float e1[3] = { ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2] };
                        const float e2[3] = { tb[0] - tc[0], tb[1] - tc[1], tb[2] - tc[2] };
                        tn[0] = e1[1] * e2[2] - e1[2] * e2[1];
                        tn[1] = e1[2] * e2[0] - e1[0] * e2[2];
                        tn[2] = e1[0] * e2[1] - e1[1] * e2[0];
                    }
                } else {                                             // vertex mode: target vertex `idx`
                    if ((uint32_t)__float_as_int(wrec.w) == idx) { qx = wrec.x; qy = wrec.y; qz = wrec.z; }
                    else {
                        qx = tgt_xyz[3ll * idx]; qy = tgt_xyz[3ll * idx + 1]; qz = tgt_xyz[3ll * idx + 2];
                        if (win) win[i] = make_float4(qx, qy, qz, __int_as_float((int)idx));
                    }
                    if (nrm.src_n) { tn[0] = nrm.tgt_n[3ll * idx]; tn[1] = nrm.tgt_n[3ll * idx + 1]; tn[2] = nrm.tgt_n[3ll * idx + 2]; }
                }
                valid = pair_eval(st, cx, cy, cz, qx, qy, qz, nrm, i, tn, thresh, bx, by, bz, dist);
            }
            if (EMIT) {
                const long long o = out.perm ? out.perm[i] : i;
                out.valid[o] = valid ? 1 : 0;
                out.b[3 * o] = bx; out.b[3 * o + 1] = by; out.b[3 * o + 2] = bz;
                out.dist[o] = dist;
                if (out.nn_idx) { out.nn_idx[o] = (int)idx; out.nn_d2[o] = __uint_as_float((uint32_t)(key >> 32)); }
            }
            if (valid) pair_add(acc, p.x, p.y, p.z, bx, by, bz, dist, pvx, pvy, pvz, d_pivot);
        }
    }
    block_store_partial(acc, red, partials + (long long)blockIdx.x * NSUMS);
}

// The stand-alone form of what k_nn_search_grid<L, true> does in its epilogue: one thread per (source slot, lane of the
// query), the same pair test, the same reduction tree, the same rows -- BIT FOR BIT.  The loop switches between "the grid
// search finishes and accumulates everything itself" and "grid search -> tree search of the hand-over list -> this kernel"
// from one iteration to the next on what the host last heard about the list (oa_icp.hip: adaptive path), and that choice
// depends on timing: it must not show in the sums.  The brute-force searches use it too (for shards whose rows fit), so
// every search mode still ends with bitwise the same matrices.
// Launched with the workgroup size the accumulating grid search of this shard uses (256 or 512 threads, oa_icp.hip:
// canon_threads): the same workgroups, hence the same rows.
constexpr int CANON_THREADS = 512;
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(CANON_THREADS) void k_pair_accumulate_canon(const DevState *__restrict__ st, const float4 *__restrict__ src4,
                                                               int ns, int L, const float *__restrict__ tgt_xyz,
                                                               unsigned long long *__restrict__ keys, int *__restrict__ prev,
                                                               float4 *__restrict__ win, const float4 *__restrict__ tri9,
                                                               NormalTest nrm, double *__restrict__ partials,
                                                               unsigned long long *__restrict__ t_acc_start)
{
    __shared__ double red[CANON_THREADS / 64][NSUMS];
    if (t_acc_start && blockIdx.x == 0 && threadIdx.x == 0) *t_acc_start = wall_clock64();   // ~ the end of the search
    if (st->halt) return;
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gt / L;
    const bool mine = (gt - i * L) == 0 && i < ns;
    bool valid = false;
    float bx = 0.f, by = 0.f, bz = 0.f;
    double dist = 0.0;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mine) {
        const unsigned long long key = keys[i];
        keys[i] = KEY_EMPTY;                                       // ready for the next iteration's atomicMin
        const uint32_t idx = (uint32_t)key;
        if (prev) prev[i] = (idx == IDX_NONE) ? -1 : (int)idx;
        p = src4[i];
        float4 wrec = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        if (win) wrec = win[i];
        if (idx != IDX_NONE) {
            float cx, cy, cz;
            co_find(st, p.x, p.y, p.z, cx, cy, cz);           // co_find                   (general.py:287)
            float qx, qy, qz;
            float tn[3] = { 0.f, 0.f, 0.f };
            if (tri9) {
                float ta[3], tb[3], tc[3], rr[3];
                const float cf[3] = { cx, cy, cz };
                load_tri(tri9, idx, ta, tb, tc);
                closest_on_tri(cf, ta, tb, tc, rr);
                qx = rr[0]; qy = rr[1]; qz = rr[2];
                if (nrm.src_n) {
                    const float e1[3] = { ta[0] - tb[0], ta[1] - tb[1], ta[2] - tb[2] };
                    const float e2[3] = { tb[0] - tc[0], tb[1] - tc[1], tb[2] - tc[2] };
                    tn[0] = e1[1] * e2[2] - e1[2] * e2[1];
                    tn[1] = e1[2] * e2[0] - e1[0] * e2[2];
                    tn[2] = e1[0] * e2[1] - e1[1] * e2[0];
                }
            } else {
                if ((uint32_t)__float_as_int(wrec.w) == idx) { qx = wrec.x; qy = wrec.y; qz = wrec.z; }
                else {
                    qx = tgt_xyz[3ll * idx]; qy = tgt_xyz[3ll * idx + 1]; qz = tgt_xyz[3ll * idx + 2];
                    if (win) win[i] = make_float4(qx, qy, qz, __int_as_float((int)idx));
                }
                if (nrm.src_n) { tn[0] = nrm.tgt_n[3ll * idx]; tn[1] = nrm.tgt_n[3ll * idx + 1]; tn[2] = nrm.tgt_n[3ll * idx + 2]; }
            }
            valid = pair_eval(st, cx, cy, cz, qx, qy, qz, nrm, i, tn, st->thresh, bx, by, bz, dist);
        }
    }
    const double pvx = st->pivot[0], pvy = st->pivot[1], pvz = st->pivot[2];
    block_store_pair(valid, (double)p.x - pvx, (double)p.y - pvy, (double)p.z - pvz, (double)bx - pvx, (double)by - pvy,
                     (double)bz - pvz, dist - st->d_pivot, red, partials + (long long)blockIdx.x * NSUMS);
}
#endif  // !OA_FAMILY_TU

// ---- fixed-order reduction of the rows of per-workgroup partials (bitwise reproducible, no float atomics) -------------
// A row is NSUMS doubles = 12 x 16 bytes.  1024 threads = 85 slices of 12 threads: slice s adds rows s, s + 85, ... in
// order (thread c of a slice owns columns 2c, 2c + 1 and loads them as one 16-byte word, 16 rows in flight per thread),
// then the slices are added in order.  Round 2's version (32 slices of 32 threads, 8 rows in flight) was sized for the
// <= 512 rows of k_pair_accumulate; the search kernels that accumulate in their epilogue write one row per workgroup,
// up to 4096 of them, and a chain of 16 round trips per thread would cost more than the launch the fusion saves.
constexpr int RED_THREADS = 1024;
constexpr int RED_SLICES = RED_THREADS / 12;        // 85

// rows [0, n_rows) of `rows` -> out[0 .. NSUMS) (shared or global).  Called by all RED_THREADS threads of a workgroup.
__device__ __forceinline__ void reduce_rows_block(const double *__restrict__ rows, int n_rows, double *out)
{
    __shared__ double red[RED_SLICES][NSUMS];
    const int s = threadIdx.x / 12, c = threadIdx.x - 12 * s;
    if (s < RED_SLICES) {
        double v0 = 0.0, v1 = 0.0;
        const double2 *__restrict__ col = (const double2 *)rows + c;       // row r, columns 2c, 2c+1: col[12 r]
        int r = s;
        for (; r + 15 * RED_SLICES < n_rows; r += 16 * RED_SLICES) {
            double2 p[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) p[u] = col[12ll * (r + RED_SLICES * u)];
#pragma unroll
            for (int u = 0; u < 16; ++u) { v0 += p[u].x; v1 += p[u].y; }
        }
        // the last (or only) batch, fewer than 16 rows: all loads at once, predicated -- ONE round trip where batches of four and
        // single rows took up to five (1954 rows at 1M points: 23 per slice = 16 + 4 + 1 + 1 + 1); same additions, same order.
        // (Four wide when that covers it: a few hundred rows are a 2562-point problem, where every instruction shows.)
        if (r + 4 * RED_SLICES < n_rows) {
            double2 p[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = r + RED_SLICES * u;
                p[u] = make_double2(0.0, 0.0);
                if (rr < n_rows) p[u] = col[12ll * rr];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r + RED_SLICES * u < n_rows) { v0 += p[u].x; v1 += p[u].y; }
        } else if (r < n_rows) {
            double2 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int rr = r + RED_SLICES * u;
                p[u] = make_double2(0.0, 0.0);
                if (rr < n_rows) p[u] = col[12ll * rr];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + RED_SLICES * u < n_rows) { v0 += p[u].x; v1 += p[u].y; }
        }
        red[s][2 * c] = v0; red[s][2 * c + 1] = v1;
    }
    __syncthreads();
    if (threadIdx.x < NSUMS) {
        double t = red[0][threadIdx.x];
        for (int k = 1; k < RED_SLICES; ++k) t += red[k][threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// How many rows a reduce launch has to add up.  Shards in the zone where the tree and the grid search take turns
// (DevState::tree_turn) have both searches enqueued and only one of them writes rows: the count is picked on the device.
struct RowSel {
    int n, n_tree;      // rows (of the grid search / the only search); rows the whole-tree search writes
    int by_turn;        // 1: n_tree when DevState::tree_turn, n otherwise
};
__device__ __forceinline__ int rows_of(const RowSel &sel, const DevState *st)
{
    return (sel.by_turn && st->tree_turn != 0) ? sel.n_tree : sel.n;
}

// stamp: where the launch leaves wall_clock64() at its start (the end of the search + accumulate part of the iteration,
// DevState::t_acc_start), or nullptr
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(RED_THREADS) void k_reduce_partials(const DevState *__restrict__ st, const double *__restrict__ partials,
                                                                 RowSel sel, double *__restrict__ sums_out,
                                                                 unsigned long long *__restrict__ stamp)
{
    if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
    reduce_rows_block(partials, st ? rows_of(sel, st) : sel.n, sums_out);
    // what follows this launch in the stream is the exchange of the sums (ncclAllReduce): the host's watchdog times THAT wait --
    // "entered the collective of iteration n + 1" here, "iteration n + 1 done" (host_halt[1]) from the solve behind it -- not
    // the search in front of it, which may take longer than any time limit without anything being wrong
    if (st && st->host_halt && threadIdx.x == 0) st->host_halt[5] = st->n + 1;
    if (st && threadIdx.x == 0) const_cast<DevState *>(st)->t_xchg_start = wall_clock64();
}

// sums over explicit pairs (contract 2: oa_kabsch).  A, B: 3 x K row-major with leading dimension ld.
__global__ __launch_bounds__(ACC_THREADS) void k_accumulate_pairs(const double *__restrict__ A,
                                                                  const double *__restrict__ B, long long K,
                                                                  long long ld, double pvx, double pvy, double pvz,
                                                                  double *__restrict__ partials)
{
    __shared__ double red[ACC_THREADS / 64][NSUMS];
    double acc[NSUMS];
#pragma unroll
    for (int k = 0; k < NSUMS; ++k) acc[k] = 0.0;
    for (long long i = (long long)blockIdx.x * ACC_THREADS + threadIdx.x; i < K; i += (long long)gridDim.x * ACC_THREADS) {
        const double a0 = A[i] - pvx, a1 = A[ld + i] - pvy, a2 = A[2 * ld + i] - pvz;
        const double b0 = B[i] - pvx, b1 = B[ld + i] - pvy, b2 = B[2 * ld + i] - pvz;
        acc[S_A] += a0; acc[S_A + 1] += a1; acc[S_A + 2] += a2;
        acc[S_B] += b0; acc[S_B + 1] += b1; acc[S_B + 2] += b2;
        acc[S_H + 0] += b0 * a0; acc[S_H + 1] += b0 * a1; acc[S_H + 2] += b0 * a2;
        acc[S_H + 3] += b1 * a0; acc[S_H + 4] += b1 * a1; acc[S_H + 5] += b1 * a2;
        acc[S_H + 6] += b2 * a0; acc[S_H + 7] += b2 * a1; acc[S_H + 8] += b2 * a2;
        acc[S_AA] += (a0 * a0 + a1 * a1) + a2 * a2;
        acc[S_BB] += (b0 * b0 + b1 * b1) + b2 * b2;
        acc[S_K] += 1.0;
    }
    block_store_partial(acc, red, partials + (long long)blockIdx.x * NSUMS);
}

// one thread: solve only (oa_kabsch / oa_kabsch_from_sums).  out[0..15] = M, out[16] = ok flag
__global__ void k_solve_only(const double *__restrict__ sums, double pvx, double pvy, double pvz, int with_scale,
                             double *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[NSUMS], M[16];
    for (int k = 0; k < NSUMS; ++k) s[k] = sums[k];
    const double pv[3] = { pvx, pvy, pvz };
    const bool ok = solve_from_sums(s, pv, (with_scale & 1) != 0, M, nullptr, false, (with_scale & 2) != 0);   // bit 1: Horn's quaternion branch
    for (int k = 0; k < 16; ++k) out[k] = ok ? M[k] : 0.0;
    out[16] = ok ? 1.0 : 0.0;
}
#endif  // !OA_FAMILY_TU

// ------------------------------------------------------------------------------------------------
// k_solve_update : operators/icp_align.py:106-149 (solve_update_block: two waves)
// ------------------------------------------------------------------------------------------------
// element `lane` (0..15) of m4_inverted(Af): the same sub-determinants, the same numerators, the same IEEE division as
// m4_inverted -- hence the same float -- but ONE division per lane instead of sixteen on one lane (they were a third of the
// post-solve chain).  Called by (at least) 16 lanes with the same Af; false if singular.
__device__ __forceinline__ bool m4_inverted_lane(const float *Af, int lane, float &out)
{
    float R[16], det;
    m4_adjoint_det(Af, R, det);
    if (det == 0.0f) return false;
    // element `lane` = (row j, column i) of the inverse = R[i][j] / det: a select chain instead of a dynamic index (registers)
    const int src = (lane & 3) * 4 + (lane >> 2);
    float nk = R[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) nk = (src == k) ? R[k] : nk;
    out = nk / det;
    return true;
}

// What follows the sums of an iteration (operators/icp_align.py:106-149), by a WORKGROUP of at least two waves; `sums`
// readable by all of them.  Called by ALL threads of the workgroup (it contains workgroup barriers); waves 2.. only pass.
//   wave 0, every lane redundantly (one lane's latency): the Kabsch solve (Jacobi, warm-started)           -> M, new_mat
//   then wave 0: matrix_world @ new_mat (lane k = element k), its inverse (lane k = element k: one division per lane),
//                |translation|, the convergence ring, n, halt, the report for the host
//        wave 1, at the same time: rotation angle (acos), d_stats (sqrt), the step record (lane k = element k of M and
//                new_mat), the angle ring, whose turn the next search is, d_pivot
// Round 2 ran all of it on one lane, one after the other: the part after the solve -- 64 products accumulated in double,
// 16 divisions, an acos, 40 stores -- took as long as the solve.
// snapshot of the loop state in LDS: every kernel that ends in solve_update_block takes it with one coalesced copy at its
// very start (the row reduction, or the wait for the mailboxes, runs meanwhile), so that the solve's reads of the state --
// three dependent global round trips on its critical path before -- are LDS reads.  Called by ALL threads; the caller's next
// __syncthreads() makes it visible.
__device__ __forceinline__ void snapshot_state(const DevState *__restrict__ st, DevState *sh)
{
    static_assert(sizeof(DevState) % 4 == 0, "DevState is copied word by word");
    for (int t = threadIdx.x; t < (int)(sizeof(DevState) / 4); t += blockDim.x) ((uint32_t *)sh)[t] = ((const uint32_t *)st)[t];
}

__device__ __forceinline__ void solve_update_block(DevState *__restrict__ st, const DevState *cs, const double *sums,
                                                   StepRecord *__restrict__ hist, int *__restrict__ todo_count,
                                                   unsigned long long t_sums_in_hand = 0ull)
{
    // st: the loop state in global memory (written); cs: its snapshot from the start of this launch (read)
    __shared__ double sh_M[16];
    __shared__ float sh_new[16], sh_mw[16];
    __shared__ int sh_go;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) {
        // the grid search's hand-over list restarts empty; what it held -- entries, and the most any one wave handed over --
        // goes to the host, which decides from it whether the next grid search can finish its leftovers itself
        const int todo_n = todo_count ? todo_count[0] : 0, todo_wave_max = todo_count ? todo_count[1] : 0;
        if (todo_count) { todo_count[0] = 0; todo_count[1] = 0; }
        if (cs->host_halt) { cs->host_halt[2] = todo_n; cs->host_halt[3] = todo_wave_max; }
        sh_go = cs->halt ? 0 : 1;
    }
    __syncthreads();
    if (!sh_go) return;                                             // (uniform)
    if (wave == 0) {
        double s[NSUMS], M[16];
        for (int k = 0; k < NSUMS; ++k) s[k] = sums[k];
        double jv[9];
        const bool jv_valid = cs->jac_valid != 0;
        for (int k = 0; k < 9; ++k) jv[k] = jv_valid ? cs->jac_v[k] : 0.0;
        const bool ok = solve_from_sums(s, cs->pivot, cs->with_scale != 0, M, jv, jv_valid);
        if (lane == 0) {
            if (!ok) {                                              // K < 3 -> ValueError in the reference: OA_E_TOO_FEW_PAIRS
                st->status = -3; st->halt = 1; sh_go = 0;
                if (cs->host_halt) cs->host_halt[0] = 1;            // the enqueuing host stops here too (it would wait for progress that never comes)
            }
            else {
                for (int k = 0; k < 9; ++k) st->jac_v[k] = jv[k];
                st->jac_valid = 1;
                for (int k = 0; k < 16; ++k) { sh_M[k] = M[k]; sh_new[k] = (float)M[k]; }   // new_mat[y][z] = M[y][z]  (:116-119)
            }
        }
    }
    __syncthreads();
    if (!sh_go) return;
    const int n = cs->n;
    if (wave == 0) {
        // matrix_world @ new_mat (:121): element `lane`, the operation order of m4_mul_m4
        if (lane < 16) {
            const int i = lane >> 2, j = lane & 3;
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) { const float p = cs->mx1[4 * i + k] * sh_new[4 * k + j]; acc += (double)p; }
            sh_mw[lane] = (float)acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // this wave's LDS writes above, its LDS reads below
        float mw[16];
        for (int k = 0; k < 16; ++k) mw[k] = sh_mw[k];
        if (lane < 16) st->mx1[lane] = mw[lane & 15];
        float inv_k = 0.f;
        const bool nonsingular = m4_inverted_lane(mw, lane & 15, inv_k);
        if (nonsingular && lane < 16) st->imx1[lane] = inv_k;
        if (lane == 0) {
            bool halt = false;
            if (!nonsingular) { st->status = -4; halt = true; }     // next make_pairs would raise   (general.py:265)
            const double trans = v3_length(sh_new[3], sh_new[7], sh_new[11]);   // new_mat.to_translation().length (:129,:138)
            bool converged = cs->converged != 0;
            if (cs->use_target) {                                   // if d_stats:                  (:136)
                st->ring_t[n % 5] = trans;                          // conv_t_list[i] = trans.length (:137-138)
                bool all = true;
                for (int k = 0; k < 5; ++k) all = all && ((k == n % 5 ? trans : cs->ring_t[k]) < cs->target_d);   // (:141)
                if (all) { converged = true; st->converged = 1; }
            }
            st->n = n + 1;                                          // n += 1                        (:151)
            if ((converged && cs->early_exit) || n + 1 >= cs->iters) halt = true;
            if (halt) st->halt = 1;
            st->t_prev_end = wall_clock64();                        // the next search starts (about) now
            if (cs->host_halt) {                                    // progress and halt flag for the enqueuing host
                // (no fence: the host only ever looks at these two words to decide whether to enqueue more; whichever arrives
                //  first, it enqueues an iteration too many -- which returns at once -- or stops where it should.  A system-scope
                //  fence here sat at the very end of every iteration's critical path.)
                cs->host_halt[1] = n + 1;
                if (halt) cs->host_halt[0] = 1;
            }
        }
    } else if (wave == 1) {
        double M[16];
        for (int k = 0; k < 16; ++k) M[k] = sh_M[k];
        const double trans = v3_length(sh_new[3], sh_new[7], sh_new[11]);
        const double angle = rotation_angle_3x3(M);
        const double K = sums[S_K];
        const double mean_dd = sums[S_D] / K;                       // mean of (d - d_pivot)
        const double mean_d = mean_dd + cs->d_pivot;
        double var = sums[S_DD] / K - mean_dd * mean_dd;
        if (var < 0.0) var = 0.0;
        if (hist && cs->max_records > 0) {
            StepRecord &r = hist[n % cs->max_records];
            if (lane < 16) { r.M[lane] = sh_M[lane]; r.new_mat[lane] = sh_new[lane]; }
            if (lane == 0) {
                r.K = K; r.mean_d = mean_d; r.std_d = sqrt(var); r.trans = trans; r.angle = angle;
                r.search_ticks = (cs->t_acc_start > cs->t_prev_end) ? (double)(cs->t_acc_start - cs->t_prev_end) : 0.0;
                r.exchange_ticks = (t_sums_in_hand > cs->t_xchg_start && cs->t_xchg_start) ? (double)(t_sums_in_hand - cs->t_xchg_start) : 0.0;
            }
        }
        if (lane == 0) {
            if (cs->use_target) st->ring_r[n % 5] = angle;
            st->d_pivot = mean_d;                                   // next iteration sums d relative to this mean
            // whose turn is the next search (DevState::tree_turn)
            const double moved = cs->use_target ? (trans + angle * cs->turn_scale) * cs->local_per_world : 0.0;
            st->tree_turn = (moved > cs->turn_limit) ? 1 : 0;
        }
    }
}

// loop start: the first search begins (about) now
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_stamp_start(DevState *__restrict__ st)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) st->t_prev_end = wall_clock64();
}

// split-phase form (one process per GPU): the sums come back from the all-reduce
__global__ __launch_bounds__(128) void k_solve_update(DevState *__restrict__ st, const double *__restrict__ sums, StepRecord *__restrict__ hist,
                               int *__restrict__ todo_count)
{
    __shared__ DevState cs;
    const unsigned long long t_in_hand = wall_clock64();            // behind the all-reduce in the stream: the world's sums are here
    snapshot_state(st, &cs);
    __syncthreads();
    solve_update_block(st, &cs, sums, hist, todo_count, t_in_hand);
}

// single-GPU form: the fixed-order reduction and the solve in one launch (same arithmetic, one boundary less)
__global__ __launch_bounds__(RED_THREADS) void k_reduce_solve_update(DevState *__restrict__ st, const double *__restrict__ partials,
                                                                     RowSel sel, double *__restrict__ sums_out,
                                                                     StepRecord *__restrict__ hist, int *__restrict__ todo_count, int stamp)
{
    __shared__ double sums[NSUMS];
    __shared__ DevState cs;
    const unsigned long long t_start = wall_clock64();
    snapshot_state(st, &cs);
    reduce_rows_block(partials, rows_of(sel, st), sums);           // (its barrier also publishes the snapshot)
    __syncthreads();
    if (stamp && threadIdx.x == 0) cs.t_acc_start = t_start;       // the search + accumulate part ended where this launch began
    if (threadIdx.x < NSUMS && sums_out) sums_out[threadIdx.x] = sums[threadIdx.x];
    solve_update_block(st, &cs, sums, hist, todo_count);
}
#endif  // !OA_FAMILY_TU

// ------------------------------------------------------------------------------------------------
// multi-device exchange (oa_create_multi): the per-iteration all-gather of the OA_NSUMS partial sums through
// mailboxes.  Two placements (oa_icp.hip picks): DEVICE -- every rank owns an inbox in its own HBM (fine-grained,
// peer-mapped); a post is a remote write into every rank's inbox over xGMI and every rank polls local memory -- or
// HOST -- one box in pinned, portable, device-mapped host memory (fallback when peer access is not available).
// box[parity][rank] holds rank's sums of sequence number `seq` (parity = seq & 1, seq = DevState::seq_base + n + 1);
// every device adds the world's posts in RANK ORDER, so all devices solve from bitwise identical sums and end with
// identical matrices -- no broadcast.  Two parities suffice: rank A can post iteration k + 2 only after its own solve
// of iteration k + 1, which needed rank B's post of k + 1, which B enqueued after its solve of iteration k had read
// A's post of k (from B's own inbox, in the device placement).
// ------------------------------------------------------------------------------------------------
struct MailSlot {
    double sums[NSUMS];
    unsigned long long seq;          // sequence number of the sums above (0 = never written)
    unsigned long long pad[7];       // 256 B per slot: a slot never shares a line with another rank's
};
constexpr int STATUS_EXCHANGE = -9;  // OA_E_RCCL: a rank's post did not arrive in time

// test hook (OA_FAULT_STALL_RANK): the stream stops here until the host releases it -- or for max_ticks of wall_clock64 at
// most, so that no test can hang a GPU
// oa_measure_valu_ceiling: VALU_BURN_CHAINS independent v_fma_f32 chains per lane, `iters` times; workgroup 0's first wave reads
// the shader clock and the constant-rate clock at both ends (clk[0] += shader cycles, clk[1] += wall_clock64 ticks)
constexpr int VALU_BURN_CHAINS = 16;
// OP 0: v_min3_f32, the half-rate class of gfx950 (v_min_f32, v_min3_f32, v_cmp_*_f32: tools/valu_rates.hip) that a third of
// k_nn_search_sorted's instructions belong to; OP 1: v_add_f32, two sources -- one wave-instruction per SIMD every two cycles,
// the issue rate itself
template <int OP>
__global__ __launch_bounds__(256) void k_valu_burn(float *__restrict__ sink, float a, float b, int iters, unsigned long long *__restrict__ clk)
{
    float acc[VALU_BURN_CHAINS];
#pragma unroll
    for (int i = 0; i < VALU_BURN_CHAINS; ++i) acc[i] = (float)threadIdx.x + (float)i;
    const unsigned long long c0 = (unsigned long long)__builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 4) {                          // (64 instructions per trip: the loop's own scalar work stays below 3 %)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < VALU_BURN_CHAINS; ++i) {
                if (OP == 0) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
            }
        }
    }
    const unsigned long long c1 = (unsigned long long)__builtin_readcyclecounter(), w1 = wall_clock64();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < VALU_BURN_CHAINS; ++i) t += acc[i];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ void k_fault_stall(const int32_t *release, unsigned long long max_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(release, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        if (wall_clock64() - t0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// step 1 on every device: fixed-order reduction of its per-workgroup partials, posted to the mailbox(es).
// dests[0..n_dest): the mailboxes this rank writes its slot of.  Host mailbox: one (the shared pinned box).  Device
// mailboxes: one inbox per rank, each in that rank's own HBM and peer-mapped -- a post is `world` remote writes of
// 200 B that travel over xGMI (PUSH model: every rank later polls its OWN memory).  skip != 0: fault injection
// (OA_FAULT_SKIP_POST_RANK), this rank's sums never arrive and the world's gather kernels run into their time limit.
__global__ __launch_bounds__(RED_THREADS) void k_reduce_post(const DevState *__restrict__ st, const double *__restrict__ partials,
                                                             RowSel sel, MailSlot *const *__restrict__ dests, int n_dest, int rank,
                                                             int world, int skip, unsigned long long *__restrict__ stamp)
{
    if (stamp && threadIdx.x == 0) *stamp = wall_clock64();
    if (st->halt || skip) return;
    __shared__ double sums[NSUMS];
    reduce_rows_block(partials, rows_of(sel, st), sums);
    __syncthreads();
    const unsigned long long seq = st->seq_base + (unsigned long long)st->n + 1ull;
    const size_t slot_ix = (size_t)(seq & 1ull) * world + rank;
    for (int e = threadIdx.x; e < n_dest * 32; e += RED_THREADS) {   // 32 threads per destination, all destinations in flight
        const int d = e >> 5, k = e & 31;
        if (k < NSUMS) __hip_atomic_store(&dests[d][slot_ix].sums[k], sums[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    for (int d = threadIdx.x; d < n_dest; d += RED_THREADS)
        __hip_atomic_store(&dests[d][slot_ix].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) const_cast<DevState *>(st)->t_xchg_start = wall_clock64();
}

// step 2 on every device: wait for the world's posts of this iteration (in this rank's mailbox), add them in rank
// order, solve + update.  Lane r waits for rank r (world <= 64).  The wait is bounded (timeout_ticks of wall_clock64):
// a rank that never posts -- its device faulted -- ends the loop with STATUS_EXCHANGE instead of hanging this one.
__global__ __launch_bounds__(128) void k_gather_solve_update(DevState *__restrict__ st, MailSlot *box, int world,
                                                            double *__restrict__ sums_out, StepRecord *__restrict__ hist,
                                                            int *__restrict__ todo_count, unsigned long long timeout_ticks)
{
    __shared__ double sums[NSUMS];
    __shared__ int arrived;
    __shared__ DevState cs;
    snapshot_state(st, &cs);
    __syncthreads();
    const bool live = cs.halt == 0;
    if (live) {
        const unsigned long long seq = cs.seq_base + (unsigned long long)cs.n + 1ull;
        MailSlot *row = box + (size_t)(seq & 1ull) * world;
        if (threadIdx.x == 0) arrived = 1;
        __syncthreads();
        if ((int)threadIdx.x < world) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(&row[threadIdx.x].seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if (wall_clock64() - t0 > timeout_ticks) { arrived = 0; break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
        if (arrived && threadIdx.x < NSUMS) {
            double t = __hip_atomic_load(&row[0].sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int r = 1; r < world; ++r)
                t += __hip_atomic_load(&row[r].sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sums[threadIdx.x] = t;
            if (sums_out) sums_out[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (live && !arrived) {                                         // (uniform: `arrived` is shared, read after the barrier)
        if (threadIdx.x == 0) {
            if (todo_count) { todo_count[0] = 0; todo_count[1] = 0; }
            st->status = STATUS_EXCHANGE;
            st->halt = 1;
            if (st->host_halt) { st->host_halt[0] = 1; __threadfence_system(); }
        }
        return;
    }
    solve_update_block(st, &cs, sums, hist, todo_count, live ? wall_clock64() : 0ull);   // (the wait above is over: the world's sums are in hand)
}

// ------------------------------------------------------------------------------------------------
// ordered compaction for make_pairs' A, B outputs (functions/general.py:313-321)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_block_counts(const unsigned char *__restrict__ valid, int ns, int *__restrict__ counts)
{
    __shared__ int wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool v = (i < ns) && valid[i];
    const unsigned long long m = __ballot(v);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single block exclusive scan of counts[n] -> offsets[n], total in offsets[n]
__global__ __launch_bounds__(1024) void k_scan_counts(const int *__restrict__ counts, int n, long long *__restrict__ offsets)
{
    __shared__ long long wtot[16];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        long long v = (i < n) ? counts[i] : 0;
        long long x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long y = __shfl_up(x, off, 64);
            if (lane >= off) x += y;
        }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        long long wpre = 0;
        for (int w = 0; w < wave; ++w) wpre += wtot[w];
        long long tot = 0;
        for (int w = 0; w < 16; ++w) tot += wtot[w];
        const long long c = carry;
        if (i < n) offsets[i] = c + wpre + x - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}

__global__ __launch_bounds__(256) void k_scatter_pairs(const unsigned char *__restrict__ valid, int ns,
                                                       const float4 *__restrict__ src4, const float *__restrict__ b,
                                                       const long long *__restrict__ offsets, long long cap,
                                                       double *__restrict__ A, double *__restrict__ B,
                                                       const int *__restrict__ members, long long begin,
                                                       int *__restrict__ pos)
{
    // pos (optional): the pair's position in the WHOLE selection -- members[i] for a spatial shard, begin + i for a
    // contiguous one -- so that the shards of a multi-device context can be merged back into vlist order
    __shared__ int wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool v = (i < ns) && valid[i];
    const unsigned long long m = __ballot(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    if (!v) return;
    int pre = __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    const long long k = offsets[blockIdx.x] + pre;
    if (k >= cap) return;
    const float4 p = src4[i];
    A[k] = (double)p.x; A[cap + k] = (double)p.y; A[2 * cap + k] = (double)p.z;
    B[k] = (double)b[3ll * i]; B[cap + k] = (double)b[3ll * i + 1]; B[2 * cap + k] = (double)b[3ll * i + 2];
    if (pos) pos[k] = members ? members[i] : (int)(begin + i);
}
#endif  // !OA_FAMILY_TU

#endif  // __HIPCC__

}  // namespace oa
