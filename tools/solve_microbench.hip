// GPU box: cost of the one-thread Kabsch solve (solve_from_sums) and of its parts, in ns per call.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I object_alignment_amd/csrc -o tools/solve_microbench.exe tools/solve_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include "oa_kernels.hpp"

using namespace oa;

__global__ void k_time_solve(const double *sums_in, int reps, int what, double *out, unsigned long long *ticks)
{
    if (threadIdx.x != 0) return;
    double s[NSUMS];
    for (int k = 0; k < NSUMS; ++k) s[k] = sums_in[k];
    const double pv[3] = { 0.0, 0.0, 0.0 };
    double chk = 0.0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < reps; ++r) {
        s[S_H + 1] += 1e-7;                                   // a different problem every repetition
        double M[16];
        if (what == 0) {
            solve_from_sums(s, pv, false, M);
            chk += M[0] + M[5] + M[11];
        } else if (what == 1) {                               // the rotation only
            double H[9], R[9];
            for (int k = 0; k < 9; ++k) H[k] = s[S_H + k];
            rotation_from_covariance(H, R);
            chk += R[0] + R[4] + R[8];
        } else if (what == 2) {                               // 4x4 inverse (float in, float out)
            float A[16], inv[16];
            for (int k = 0; k < 16; ++k) A[k] = (k % 5 == 0) ? 1.f : 0.01f * (float)k;
            A[3] = (float)s[S_H + 1];
            m4_inverted(A, inv);
            chk += inv[0] + inv[7];
        } else {                                              // rotation angle
            for (int k = 0; k < 16; ++k) M[k] = (k % 5 == 0) ? 1.0 : 1e-3 * k;
            M[1] = s[S_H + 1] * 1e-6;
            chk += rotation_angle_3x3(M);
        }
    }
    const unsigned long long t1 = wall_clock64();
    out[0] = chk;
    ticks[0] = t1 - t0;
}

int main()
{
    // sums of a realistic iteration: 1e6 pairs, a ~ U[-1,1]^3, b = R a + t + noise
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    std::normal_distribution<double> N(0.0, 1e-3);
    double s[NSUMS] = { 0 };
    const double ang = 0.02, c = cos(ang), sn = sin(ang);
    for (int i = 0; i < 1000000; ++i) {
        const double a[3] = { U(g), U(g), U(g) };
        const double b[3] = { c * a[0] - sn * a[1] + 0.004 + N(g), sn * a[0] + c * a[1] - 0.003 + N(g), a[2] + 0.002 + N(g) };
        for (int k = 0; k < 3; ++k) { s[S_A + k] += a[k]; s[S_B + k] += b[k]; }
        for (int p = 0; p < 3; ++p) for (int q = 0; q < 3; ++q) s[S_H + 3 * p + q] += b[p] * a[q];
        s[S_AA] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        s[S_BB] += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        s[S_K] += 1.0;
    }
    {   // sweeps the Jacobi takes on this input (host run of the same code)
        double H[9], ca[3], cb[3];
        for (int i = 0; i < 3; ++i) { ca[i] = s[S_A + i] / s[S_K]; cb[i] = s[S_B + i] / s[S_K]; }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[3 * i + j] = s[S_H + 3 * i + j] - s[S_K] * cb[i] * ca[j];
        double gm[3][3], v[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gm[i][j] = H[3 * i + j];
        int rot = 0, sweeps = 0;
        for (; sweeps < 64; ++sweeps) {
            const bool a0 = jacobi_rotate<0, 1>(gm, v), a1 = jacobi_rotate<0, 2>(gm, v), a2 = jacobi_rotate<1, 2>(gm, v);
            rot += (int)a0 + (int)a1 + (int)a2;
            if (!(a0 || a1 || a2)) break;
        }
        printf("jacobi on this input: %d sweeps (incl. the final empty one), %d rotations\n", sweeps + 1, rot);
    }
    double *d_s, *d_out; unsigned long long *d_t;
    (void)hipMalloc(&d_s, sizeof s); (void)hipMalloc(&d_out, 8); (void)hipMalloc(&d_t, 8);
    (void)hipMemcpy(d_s, s, sizeof s, hipMemcpyHostToDevice);
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    const char *names[4] = { "solve_from_sums", "rotation_from_covariance", "m4_inverted", "rotation_angle_3x3" };
    for (int what = 0; what < 4; ++what) {
        const int reps = 200;
        hipLaunchKernelGGL(k_time_solve, dim3(1), dim3(64), 0, 0, d_s, reps, what, d_out, d_t);   // warm-up (code fetch)
        hipLaunchKernelGGL(k_time_solve, dim3(1), dim3(64), 0, 0, d_s, reps, what, d_out, d_t);
        (void)hipDeviceSynchronize();
        unsigned long long t; double chk;
        (void)hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&chk, d_out, 8, hipMemcpyDeviceToHost);
        printf("%-26s %8.0f ns per call   (checksum %.6f)\n", names[what], (double)t / reps * 1e6 / khz, chk);
    }
    return 0;
}
