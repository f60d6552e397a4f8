// oa_affine.hpp -- the general form of affine_matrix_from_points (functions/general.py:105-217): any ndims >= 2 (the
// reference takes any, :149-150; 2..8 run on fixed-size registers / private arrays, 9..AFF_MAXD_HEAP on the same solve with
// its matrices in a device workspace and one workgroup per row / per Gram entry for the sums), and the shear=True (full
// affine, Hartley & Zisserman) branch the reference's signature defaults to.
//
// The ICP loop itself only ever asks for the 3-D rigid / similarity solve (k_solve_update, oa_kabsch: 24 running sums
// and a 3x3 solve).  This is the rest of the contract, on the same accumulate-then-solve plan:
//   k_affine_colsums   per-workgroup column sums of v0 and v1                    -> centroids  (:160, :164)
//   k_affine_gram      per-workgroup Gram matrix of the centred, stacked points  G = [a; b] [a; b]^T  (2n x 2n)
//   k_affine_solve     one thread, fp64:
//       shear          the n dominant eigenvectors of G span the same subspace as the first n right singular
//                      vectors of [a; b]^T (:170-173); with them B = top half, C = bottom half, t = C pinv(B) (:174)
//       otherwise      H = b a^T is the lower-left block of G (:181); R = U V^T with the reflection fix on the LAST
//                      singular direction (:183-187); optional uniform scale sqrt(tr G_bb / tr G_aa) (:208-212)
//       then           M = T(c1) [t] T(-c0), normalised by its corner (:215-216)
// Reductions run in a fixed order (no float atomics): bitwise reproducible.
#pragma once
#include "oa_kernels.hpp"

namespace oa {

constexpr int AFF_MAXD = 8;                 // largest ndims of the fixed-size kernels
constexpr int AFF_MAXD_HEAP = 64;           // largest ndims at all (k_affine_solve<16 / 32 / 64, true>: matrices in a workspace)
constexpr int AFF_M2 = 2 * AFF_MAXD;        // rows of the stacked point matrix
constexpr int AFF_TILE = 128;               // columns per LDS tile of k_affine_gram

#if defined(__HIPCC__)

// partials[block][0 .. 2n): sums of the block's columns of v0 (rows 0..n-1) and v1 (rows n..2n-1)
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(256) void k_affine_colsums(const double *__restrict__ v0, const double *__restrict__ v1, int n,
                                                        long long K, long long ld, double *__restrict__ partials)
{
    __shared__ double red[4][AFF_M2];
    double acc[AFF_M2];
#pragma unroll
    for (int r = 0; r < AFF_M2; ++r) acc[r] = 0.0;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < K; c += (long long)gridDim.x * 256) {
#pragma unroll
        for (int r = 0; r < AFF_MAXD; ++r)
            if (r < n) { acc[r] += v0[(long long)r * ld + c]; acc[AFF_MAXD + r] += v1[(long long)r * ld + c]; }
    }
#pragma unroll
    for (int r = 0; r < AFF_M2; ++r) {
        const double t = wave_sum(acc[r]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][r] = t;
    }
    __syncthreads();
    if (threadIdx.x < AFF_M2) {
        const int r = threadIdx.x;
        partials[(long long)blockIdx.x * AFF_M2 + r] = ((red[0][r] + red[1][r]) + red[2][r]) + red[3][r];
    }
}

// out[j] = sum over blocks (in order) of partials[block][j], j < width
__global__ void k_affine_reduce(const double *__restrict__ partials, int n_blocks, int width, double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= width) return;
    double t = 0.0;
    for (int b = 0; b < n_blocks; ++b) t += partials[(long long)b * width + j];
    out[j] = t;
}

// partials[block][i * 2n + j] = sum over the block's columns c of x_i(c) x_j(c), x = [v0 - c0; v1 - c1] (2n rows).
// A tile of columns is staged in LDS (coalesced row reads); thread t < (2n)^2 owns entry (t / 2n, t % 2n).
__global__ __launch_bounds__(256) void k_affine_gram(const double *__restrict__ v0, const double *__restrict__ v1, int n,
                                                     long long K, long long ld, const double *__restrict__ colsums,
                                                     long long cols_per_block, double *__restrict__ partials)
{
    __shared__ double tile[AFF_M2][AFF_TILE];
    const int m = 2 * n;
    const int t = threadIdx.x, ti = t / m, tj = t % m;
    const bool owner = t < m * m;
    const long long c0 = (long long)blockIdx.x * cols_per_block, c1 = (c0 + cols_per_block < K) ? c0 + cols_per_block : K;
    double acc = 0.0;
    for (long long base = c0; base < c1; base += AFF_TILE) {
        for (int e = t; e < m * AFF_TILE; e += 256) {
            const int r = e / AFF_TILE, c = e % AFF_TILE;
            const long long col = base + c;
            double v = 0.0;
            if (col < c1) {
                const double mean = (r < n ? colsums[r] : colsums[AFF_MAXD + (r - n)]) / (double)K;
                v = (r < n ? v0[(long long)r * ld + col] : v1[(long long)(r - n) * ld + col]) - mean;
            }
            tile[r][c] = v;
        }
        __syncthreads();
        if (owner)
            for (int c = 0; c < AFF_TILE; ++c) acc += tile[ti][c] * tile[tj][c];
        __syncthreads();
    }
    if (owner) partials[(long long)blockIdx.x * (AFF_M2 * AFF_M2) + t] = acc;
    else if (t < AFF_M2 * AFF_M2) partials[(long long)blockIdx.x * (AFF_M2 * AFF_M2) + t] = 0.0;
}
#endif  // !OA_FAMILY_TU

// ---- small dense fp64 routines (one thread; sizes <= 16) ---------------------------------------------------------------
// cyclic Jacobi on a symmetric m x m matrix: A -> diagonal (eigenvalues), V = eigenvectors in columns
template <int LD>
__device__ inline void aff_jacobi_eig(int m, double (*A)[LD], double (*V)[LD])
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < m; ++i) {
            diag += A[i][i] * A[i][i];
            for (int j = i + 1; j < m; ++j) off += A[i][j] * A[i][j];
        }
        if (!(off > 1e-33 * diag) || off == 0.0) break;
        for (int p = 0; p < m - 1; ++p)
            for (int q = p + 1; q < m; ++q) {
                const double apq = A[p][q];
                if (apq == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int k = 0; k < m; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = cs * a - sn * b; A[k][q] = sn * a + cs * b; }
                for (int k = 0; k < m; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = cs * a - sn * b; A[q][k] = sn * a + cs * b; }
                for (int k = 0; k < m; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
            }
    }
}

// one-sided Jacobi SVD of an n x n matrix: on return G = H V (columns u_j s_j), V orthogonal, sg[j] = |column j of G|,
// order[] = column indices by descending singular value
template <int LD>
__device__ inline void aff_svd(int n, double (*G)[LD], double (*V)[LD], double *sg, int *order)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool any = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double npp = 0.0, nqq = 0.0, dpq = 0.0;
                for (int k = 0; k < n; ++k) { npp += G[k][p] * G[k][p]; nqq += G[k][q] * G[k][q]; dpq += G[k][p] * G[k][q]; }
                if (dpq == 0.0 || dpq * dpq <= 1e-30 * (npp * nqq)) continue;
                any = true;
                const double w = nqq - npp, d2 = 2.0 * dpq;
                const double tt = (w >= 0.0 ? d2 : -d2) / (fabs(w) + sqrt(w * w + d2 * d2));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int k = 0; k < n; ++k) {
                    const double gp = G[k][p], gq = G[k][q], vp = V[k][p], vq = V[k][q];
                    G[k][p] = cs * gp - sn * gq; G[k][q] = sn * gp + cs * gq;
                    V[k][p] = cs * vp - sn * vq; V[k][q] = sn * vp + cs * vq;
                }
            }
        if (!any) break;
    }
    for (int j = 0; j < n; ++j) {
        double s2 = 0.0;
        for (int k = 0; k < n; ++k) s2 += G[k][j] * G[k][j];
        sg[j] = sqrt(s2);
        order[j] = j;
    }
    for (int a = 1; a < n; ++a) {                                   // insertion sort, descending, stable
        const int oj = order[a];
        int b = a - 1;
        while (b >= 0 && sg[order[b]] < sg[oj]) { order[b + 1] = order[b]; --b; }
        order[b + 1] = oj;
    }
}

// (a: n x n scratch, overwritten)
template <int LD>
__device__ inline double aff_det(int n, double (*R)[LD], double (*a)[LD])
{
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) a[i][j] = R[i][j];
    double det = 1.0;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return 0.0;
        if (piv != c) { for (int k = 0; k < n; ++k) { const double tmp = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = tmp; } det = -det; }
        det *= a[c][c];
        for (int r = c + 1; r < n; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int k = c; k < n; ++k) a[r][k] -= f * a[c][k];
        }
    }
    return det;
}

// out[0 .. (n+1)^2) = M (row-major), out[(n+1)^2] = 1 when K >= ndims (else 0: the reference's ValueError)
// D = the capacity the arrays are laid out for (colsums: v1's sums start at D; n <= D).  HEAP: every matrix lives in `ws`
// (aff_ws_doubles(D) doubles of device memory) instead of the thread's private arrays -- 64 dimensions would be 600 KB of
// them.  The arithmetic is the same code either way.
constexpr size_t aff_ws_doubles(int D) { return (size_t)2 * (2 * D) * (2 * D) + (size_t)10 * D * D + (size_t)8 * D + 64; }
#define AFF_MAT(name, R, Cc) double name##_loc[HEAP ? 1 : (R)][HEAP ? 1 : (Cc)]; \
                             double (*name)[Cc] = HEAP ? (double (*)[Cc])(ws_at += (size_t)(R) * (Cc), ws_at - (size_t)(R) * (Cc)) : (double (*)[Cc])&name##_loc[0][0]
#define AFF_VEC(type, name, N) type name##_loc[HEAP ? 1 : (N)]; \
                               type *name = HEAP ? (type *)(ws_at += (N), ws_at - (N)) : &name##_loc[0]
template <int D, bool HEAP>
__global__ void k_affine_solve(const double *__restrict__ colsums, const double *__restrict__ gram, int n, long long K,
                               int shear, int with_scale, double *__restrict__ out, double *__restrict__ ws)
{
    constexpr int AFF_MAXD = D, AFF_M2 = 2 * D;                     // (shadow the fixed-size constants: the body below is written in them)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double *ws_at = ws;
    const int m = 2 * n, w = n + 1;
    if (K < n) { out[w * w] = 0.0; return; }
    AFF_VEC(double, c0, AFF_MAXD); AFF_VEC(double, c1, AFF_MAXD);
    AFF_MAT(T, AFF_MAXD, AFF_MAXD); AFF_MAT(det_tmp, AFF_MAXD, AFF_MAXD);
    for (int i = 0; i < n; ++i) { c0[i] = colsums[i] / (double)K; c1[i] = colsums[AFF_MAXD + i] / (double)K; }
    if (shear) {
        AFF_MAT(A, AFF_M2, AFF_M2); AFF_MAT(V, AFF_M2, AFF_M2);
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) A[i][j] = 0.5 * (gram[i * m + j] + gram[j * m + i]);
        aff_jacobi_eig(m, A, V);
        AFF_VEC(double, ord_d, AFF_M2);                             // (indices kept as doubles in the workspace: one element type)
        int *ord = (int *)ord_d;
        for (int j = 0; j < m; ++j) ord[j] = j;
        for (int a = 1; a < m; ++a) {                               // eigenvalues descending
            const int oj = ord[a];
            int b = a - 1;
            while (b >= 0 && A[ord[b]][ord[b]] < A[oj][oj]) { ord[b + 1] = ord[b]; --b; }
            ord[b + 1] = oj;
        }
        // B = rows 0..n-1, C = rows n..2n-1 of the n dominant eigenvectors (:172-173); t = C pinv(B) (:174)
        AFF_MAT(B, AFF_MAXD, AFF_MAXD); AFF_MAT(C, AFF_MAXD, AFF_MAXD); AFF_MAT(Gs, AFF_MAXD, AFF_MAXD); AFF_MAT(Vs, AFF_MAXD, AFF_MAXD);
        AFF_VEC(double, sg, AFF_MAXD); AFF_VEC(double, so_d, AFF_MAXD);
        int *so = (int *)so_d;
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < n; ++k) { B[i][k] = V[i][ord[k]]; C[i][k] = V[n + i][ord[k]]; Gs[i][k] = B[i][k]; }
        aff_svd(n, Gs, Vs, sg, so);                                 // B Vs = U S  =>  pinv(B) = Vs S^+ U^T
        const double cut = 1e-15 * sg[so[0]];                       // numpy.linalg.pinv's default rcond
        AFF_MAT(P, AFF_MAXD, AFF_MAXD);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int k = 0; k < n; ++k)
                    if (sg[k] > cut) acc += Vs[i][k] * Gs[j][k] / (sg[k] * sg[k]);     // Vs[i][k] (1 / s_k) u_k[j], u_k = Gs[:, k] / s_k
                P[i][j] = acc;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc += C[i][k] * P[k][j];
                T[i][j] = acc;
            }
    } else {
        AFF_MAT(H, AFF_MAXD, AFF_MAXD); AFF_MAT(Vs, AFF_MAXD, AFF_MAXD); AFF_MAT(U, AFF_MAXD, AFF_MAXD);
        AFF_VEC(double, sg, AFF_MAXD); AFF_VEC(double, so_d, AFF_MAXD);
        int *so = (int *)so_d;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i][j] = gram[(n + i) * m + j];    // dot(v1c, v0c.T)  (:181)
        aff_svd(n, H, Vs, sg, so);
        // U columns in descending order; zero singular directions are completed to an orthonormal basis
        for (int k = 0; k < n; ++k) {
            const int col = so[k];
            double nn = 0.0;
            if (sg[col] > 1e-300 && sg[col] > 1e-14 * sg[so[0]]) {
                for (int r = 0; r < n; ++r) U[r][k] = H[r][col] / sg[col];
            } else {
                for (int e = 0; e < n && !(nn > 0.25); ++e) {       // Gram-Schmidt on the unit vectors
                    for (int r = 0; r < n; ++r) U[r][k] = r == e ? 1.0 : 0.0;
                    for (int j = 0; j < k; ++j) {
                        double d = 0.0;
                        for (int r = 0; r < n; ++r) d += U[r][k] * U[r][j];
                        for (int r = 0; r < n; ++r) U[r][k] -= d * U[r][j];
                    }
                    nn = 0.0;
                    for (int r = 0; r < n; ++r) nn += U[r][k] * U[r][k];
                }
                nn = sqrt(nn);
                for (int r = 0; r < n; ++r) U[r][k] /= nn;
            }
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double acc = 0.0;
                for (int k = 0; k < n; ++k) acc += U[i][k] * Vs[j][so[k]];                           // R = u vh  (:183)
                T[i][j] = acc;
            }
        if (aff_det(n, T, det_tmp) < 0.0) {                                  // not a right-handed system (:184-187)
            const int last = so[n - 1];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) T[i][j] -= 2.0 * U[i][n - 1] * Vs[j][last];
        }
        if (with_scale) {                                           // :208-212
            double s0 = 0.0, s1 = 0.0;
            for (int i = 0; i < n; ++i) { s0 += gram[i * m + i]; s1 += gram[(n + i) * m + (n + i)]; }
            const double sc = sqrt(s1 / s0);
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) T[i][j] *= sc;
        }
    }
    // M = inv(M1) (M M0): linear part T, translation c1 - T c0 (:215); M /= M[n][n] (:216) is a division by 1
    for (int i = 0; i < n; ++i) {
        double tr = c1[i];
        for (int j = 0; j < n; ++j) { out[i * w + j] = T[i][j]; tr -= T[i][j] * c0[j]; }
        out[i * w + n] = tr;
    }
    for (int j = 0; j < n; ++j) out[n * w + j] = 0.0;
    out[n * w + n] = 1.0;
    out[w * w] = 1.0;
}
#undef AFF_MAT
#undef AFF_VEC

// ndims > 8: the same sums with one workgroup per row / per Gram entry (fixed order inside a workgroup: reproducible); the
// layouts are those of the fixed-size kernels with D in place of AFF_MAXD.
// colsums[r < n ? r : D + (r - n)] = sum over the columns of row r of [v0; v1]
#if !defined(OA_FAMILY_TU)      // plain kernels are compiled once, in the host translation unit (oa_icp.hip)
__global__ __launch_bounds__(256) void k_affine_rowsum_any(const double *__restrict__ v0, const double *__restrict__ v1, int n, int D,
                                                           long long K, long long ld, double *__restrict__ colsums)
{
    __shared__ double red[4];
    const int r = blockIdx.x;
    const double *row = r < n ? v0 + (long long)r * ld : v1 + (long long)(r - n) * ld;
    double acc = 0.0;
    for (long long c = threadIdx.x; c < K; c += 256) acc += row[c];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) colsums[r < n ? r : D + (r - n)] = ((red[0] + red[1]) + red[2]) + red[3];
}

// gram[i * 2n + j] = sum over the columns of x_i x_j, x = [v0 - c0; v1 - c1]; workgroup (i, j) = blockIdx.x / 2n, % 2n
__global__ __launch_bounds__(256) void k_affine_gram_any(const double *__restrict__ v0, const double *__restrict__ v1, int n, int D,
                                                         long long K, long long ld, const double *__restrict__ colsums,
                                                         double *__restrict__ gram)
{
    __shared__ double red[4];
    const int m = 2 * n, i = blockIdx.x / m, j = blockIdx.x % m;
    const double *ri = i < n ? v0 + (long long)i * ld : v1 + (long long)(i - n) * ld;
    const double *rj = j < n ? v0 + (long long)j * ld : v1 + (long long)(j - n) * ld;
    const double mi = (i < n ? colsums[i] : colsums[D + (i - n)]) / (double)K, mj = (j < n ? colsums[j] : colsums[D + (j - n)]) / (double)K;
    double acc = 0.0;
    for (long long c = threadIdx.x; c < K; c += 256) acc += (ri[c] - mi) * (rj[c] - mj);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) gram[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
#endif  // !OA_FAMILY_TU

#endif  // __HIPCC__
}  // namespace oa
