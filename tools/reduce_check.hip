// reduce_check.hip -- wave_reduce_scatter (oa_kernels.hpp: v_permlane32_swap / v_permlane16_swap / DPP) against the form it
// replaced (ds_bpermute shuffles with selects), bit for bit, on random doubles of mixed magnitude and sign.
// build: __graft_entry__.build_tools()     run on the GPU box: tools/reduce_check.exe [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "oa_kernels.hpp"

namespace old_form {
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], int lane, double *red_row)
{
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
    double u[N / 2], w[3];
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
        const double keep = b5 ? v[N / 2 + k] : v[k], send = b5 ? v[k] : v[N / 2 + k];
        u[k] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < N / 4; ++k) {
        const double keep = b4 ? u[N / 4 + k] : u[k], send = b4 ? u[k] : u[N / 4 + k];
        w[k] = keep + __shfl_xor(send, 16, 64);
    }
    if (N == 12) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double t = w[k];
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
            w[k] = t;
        }
        if ((lane & 15) == 0) {
            const int first = (b5 ? 6 : 0) + (b4 ? 3 : 0);
            red_row[first] = w[0]; red_row[first + 1] = w[1]; red_row[first + 2] = w[2];
        }
    } else {
        const double keep = b3 ? w[1] : w[0], send = b3 ? w[0] : w[1];
        double t = keep + __shfl_xor(send, 8, 64);
        t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
        if ((lane & 7) == 0) red_row[(b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0)] = t;
    }
}
}  // namespace old_form

template <int N, bool OLD>
__global__ void k_reduce(const double *__restrict__ in, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    double v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = in[(wave * 64 + lane) * N + k];
    if (OLD) old_form::wave_reduce_scatter<N>(v, lane, out + wave * N);
    else oa::wave_reduce_scatter<N>(v, lane, out + wave * N);
}

template <int N>
static long long check(int waves, unsigned seed)
{
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> mant(-1.0, 1.0);
    std::uniform_int_distribution<int> expo(-30, 30);
    std::vector<double> h((size_t)waves * 64 * N);
    for (auto &x : h) x = ldexp(mant(rng), expo(rng));
    double *d_in, *d_a, *d_b;
    hipMalloc(&d_in, h.size() * 8); hipMalloc(&d_a, (size_t)waves * N * 8); hipMalloc(&d_b, (size_t)waves * N * 8);
    hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_reduce<N, true>), dim3(waves / 4), dim3(256), 0, 0, d_in, d_a);
    hipLaunchKernelGGL((k_reduce<N, false>), dim3(waves / 4), dim3(256), 0, 0, d_in, d_b);
    std::vector<double> a((size_t)waves * N), b(a.size());
    hipMemcpy(a.data(), d_a, a.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d_b, b.size() * 8, hipMemcpyDeviceToHost);
    long long bad = 0;
    for (size_t i = 0; i < a.size(); ++i) if (memcmp(&a[i], &b[i], 8) != 0) ++bad;
    // and against a plain host sum, loosely (catches a wrong lane pattern that both forms might share)
    double worst = 0.0;
    for (int w = 0; w < waves; ++w)
        for (int k = 0; k < N; ++k) {
            long double s = 0.0L, m = 0.0L;
            for (int l = 0; l < 64; ++l) { const double x = h[((size_t)w * 64 + l) * N + k]; s += x; m += fabsl(x); }
            const double e = (double)(fabsl(s - (long double)b[(size_t)w * N + k]) / (m + 1e-300L));
            if (e > worst) worst = e;
        }
    printf("N = %2d: %d waves, %lld of %zu sums differ bitwise from the shuffle form; worst |error| / sum|x| against a long double host sum %.2e\n",
           N, waves, bad, a.size(), worst);
    hipFree(d_in); hipFree(d_a); hipFree(d_b);
    return bad + (worst > 1e-14 ? 1 : 0);
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) & ~3 : 4096;
    long long bad = check<12>(waves, 1u) + check<8>(waves, 2u);
    printf(bad ? "MISMATCH\n" : "identical\n");
    return bad ? 1 : 0;
}
