// mfma_microbench.hip -- probes for the OA_NN_MFMA experiment (a conservative pre-filter of the brute-force nearest-
// vertex search on the matrix cores):
//   1. rate: v_mfma_f32_32x32x16_f16 with a zero accumulator followed by the sign test of its 16 results per lane
//      (8 x v_or3_b32 + compare + branch) -- cycles per 1024 point-target pairs per SIMD, the loop the search would run;
//   2. arithmetic: how far is the f32 result of one such MFMA from the exact sum of its 16 products?  (the filter's
//      threshold must cover it; printed relative to sum |a_k b_k| in units of u = 2^-24);
//   3. VALU probes the round-1 microbenchmark did not have: v_or3_b32, v_pk_fma_f16, v_pk_min_f16, v_dot2_f32_f16.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_microbench.exe mfma_microbench.hip     Run: ./mfma_microbench.exe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16_ __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int or16(const float16_ d)
{
    int r = __float_as_int(d[0]) | __float_as_int(d[1]) | __float_as_int(d[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) r = r | __float_as_int(d[k]) | __float_as_int(d[k + 1]);
    return r | __float_as_int(d[15]);
}

// rate probe: TILES target tiles x 8 point blocks per wave and outer iteration
constexpr int TILES = 16;
__global__ __launch_bounds__(256, 1) void k_rate(const half8 *__restrict__ img, int iters, int *__restrict__ out)
{
    __shared__ half8 tile[TILES * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < TILES * 64; i += 256) tile[i] = img[i];
    __syncthreads();
    half8 B[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) B[b] = img[(TILES + b) * 64 + lane];
    int flagged = 0;
    const float16_ zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < TILES; ++t) {
            const half8 a = tile[t * 64 + lane];
#pragma unroll
            for (int b = 0; b < 8; b += 2) {
                const float16_ d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[b], zero, 0, 0, 0);
                const float16_ d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[b + 1], zero, 0, 0, 0);
                const int s0 = or16(d0), s1 = or16(d1);
                if (__any(s0 < 0)) { flagged += 1; B[b][0] += (_Float16)1; }     // never taken with the probe's data (all scores > 0)
                if (__any(s1 < 0)) { flagged += 2; B[b + 1][0] += (_Float16)1; }
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = flagged;
}

// the same loop with other ways to look at the 16 results per lane: FL = 0 v_or3_b32 on the bits (8 per MFMA), 1 the
// NaN-sticky sum acc = fma(d, +inf, acc) started at +inf (16 full-rate FMAs: a negative d makes -inf, inf - inf = NaN),
// 2 half and half (8 results through 4 v_or3, 8 through 8 v_fma), 3 v_min3_f32 (8 per MFMA), 4 nothing (MFMA only)
template <int FL>
__global__ __launch_bounds__(256, 1) void k_rate_fl(const half8 *__restrict__ img, int iters, int *__restrict__ out)
{
    __shared__ half8 tile[TILES * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < TILES * 64; i += 256) tile[i] = img[i];
    __syncthreads();
    half8 B[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) B[b] = img[(TILES + b) * 64 + lane];
    int flagged = 0;
    const float16_ zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const float inf = __builtin_inff();
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < TILES; ++t) {
            const half8 a = tile[t * 64 + lane];
            int acc = 0;
            float facc = inf, macc = inf;
            float16_ d_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[0], zero, 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float16_ d = d_next;
                if (b + 1 < 8) d_next = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[b + 1], zero, 0, 0, 0);
                if (FL == 0) {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) acc = acc | __float_as_int(d[k]) | __float_as_int(d[k + 1]);
                } else if (FL == 1) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) facc = __builtin_fmaf(d[k], inf, facc);
                } else if (FL == 2) {
#pragma unroll
                    for (int k = 0; k < 8; k += 2) acc = acc | __float_as_int(d[k]) | __float_as_int(d[k + 1]);
#pragma unroll
                    for (int k = 8; k < 16; ++k) facc = __builtin_fmaf(d[k], inf, facc);
                } else if (FL == 3) {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) macc = __builtin_fminf(macc, __builtin_fminf(d[k], d[k + 1]));
                } else {
                    acc |= __float_as_int(d[0]);
                }
            }
            const bool bad = acc < 0 || !(facc == inf) || macc < 0.f;
            if (__any(bad)) { flagged += 1; B[0][0] += (_Float16)1; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = flagged;
}

// arithmetic probe: one wave, D = A x B for one tile, all 1024 results written out (row = target, col = point)
__global__ void k_one(const half8 *__restrict__ a, const half8 *__restrict__ b, float *__restrict__ d)
{
    const int lane = threadIdx.x;
    const float16_ zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const float16_ r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[lane], b[lane], zero, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        d[row * 32 + col] = r[reg];
    }
}

constexpr int ITERS = 4096, UNROLL = 16;
#define VPROBE(NAME, TYPE, INIT, ASM)                                                                \
__global__ __launch_bounds__(256) void NAME(float *out, float a, float b)                            \
{                                                                                                    \
    TYPE acc[UNROLL];                                                                                \
    const int va = __float_as_int(a) ^ threadIdx.x, vb = __float_as_int(b) + threadIdx.x;            \
    _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) acc[i] = INIT;                                \
    for (int it = 0; it < ITERS; ++it) {                                                             \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(va), "v"(vb)); \
    }                                                                                                \
    int s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s += (int)acc[i];                             \
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;                                                  \
}
VPROBE(k_or3, int, (int)threadIdx.x + i, "v_or3_b32 %0, %0, %1, %2")
VPROBE(k_fma, float, (float)threadIdx.x + i, "v_fma_f32 %0, %1, %2, %0")
VPROBE(k_pk_fma_f16, int, (int)threadIdx.x + i, "v_pk_fma_f16 %0, %1, %2, %0")
VPROBE(k_pk_min_f16, int, (int)threadIdx.x + i, "v_pk_min_f16 %0, %0, %1")
VPROBE(k_dot2_f32_f16, float, (float)threadIdx.x + i, "v_dot2_f32_f16 %0, %1, %2, %0")
VPROBE(k_cmp_lt_i32, int, (int)threadIdx.x + i, "v_cmp_lt_i32 vcc, %0, %1")

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static double h2d(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (double)h; }

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("device   CUs %d  clock %.0f MHz\n", cus, clk / 1e6);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));

    // ---- 1. rate
    {
        std::vector<uint16_t> h((TILES + 8) * 64 * 8);
        srand(7);
        for (size_t i = 0; i < h.size(); ++i) h[i] = f2h(0.25f + 0.5f * (rand() / (float)RAND_MAX));   // all products > 0: never flagged
        half8 *d_img; int *d_out;
        CHK(hipMalloc(&d_img, h.size() * 2)); CHK(hipMalloc(&d_out, sizeof(int) * 256 * cus * 16));
        CHK(hipMemcpy(d_img, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (int wpc = 4; wpc <= 16; wpc *= 2) {                   // waves per CU: 4 (one per SIMD), 8, 16
            const int blocks = cus * wpc / 4, iters = 64;
            hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, d_img, 2, d_out);
            CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(256), 0, 0, d_img, iters, d_out);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double mfma_per_simd = (double)iters * TILES * 8 * wpc / 4.0;
            const double pairs = (double)iters * TILES * 8 * 1024.0 * blocks * 4;
            printf("mfma 32x32x16 f16 + sign test, %2d waves/CU: %.3f ms  %.1f cycles per MFMA per SIMD (at max clock)  %.1f T pairs/s  -> 1e12 pairs in %.1f ms\n",
                   wpc, best, best * 1e-3 * clk / mfma_per_simd, pairs / (best * 1e-3) / 1e12, 1e12 / (pairs / (best * 1e-3)) * 1e3);
        }
        struct { const char *name; void (*k)(const half8 *, int, int *); } fl[] = {
            { "8 x v_or3_b32 (running, one test per tile)", k_rate_fl<0> }, { "16 x v_fma_f32 NaN-sticky", k_rate_fl<1> },
            { "4 x v_or3 + 8 x v_fma", k_rate_fl<2> }, { "8 x v_min3_f32", k_rate_fl<3> }, { "MFMA alone", k_rate_fl<4> } };
        for (auto &f : fl) {
            const int wpc = 16, blocks = cus * wpc / 4, iters = 64;
            hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, d_img, 2, d_out);
            CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(f.k, dim3(blocks), dim3(256), 0, 0, d_img, iters, d_out);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double mfma_per_simd = (double)iters * TILES * 8 * wpc / 4.0;
            const double pairs = (double)iters * TILES * 8 * 1024.0 * blocks * 4;
            printf("mfma + %-44s 16 waves/CU: %.1f cycles per MFMA per SIMD (at max clock)  -> 1e12 pairs in %.1f ms\n",
                   f.name, best * 1e-3 * clk / mfma_per_simd, 1e12 / (pairs / (best * 1e-3)) * 1e3);
        }
        CHK(hipFree(d_img)); CHK(hipFree(d_out));
    }

    // ---- 2. arithmetic of one MFMA: results vs the exact sum of the 16 products
    {
        std::vector<uint16_t> ha(64 * 8), hb(64 * 8);
        double worst = 0.0, worst_rel_result = 0.0;
        half8 *d_a, *d_b; float *d_d;
        CHK(hipMalloc(&d_a, 64 * 16)); CHK(hipMalloc(&d_b, 64 * 16)); CHK(hipMalloc(&d_d, 4096));
        std::vector<float> hd(1024);
        srand(11);
        for (int trial = 0; trial < 200; ++trial) {
            // magnitudes spread over many binades, mixed signs: the cancellation the filter's score has
            for (size_t i = 0; i < ha.size(); ++i) {
                const float ma = ldexpf(0.5f + 0.5f * (rand() / (float)RAND_MAX), -(rand() % 14));
                const float mb = ldexpf(0.5f + 0.5f * (rand() / (float)RAND_MAX), -(rand() % 14) + (trial % 8));
                ha[i] = f2h((rand() & 1) ? ma : -ma);
                hb[i] = f2h((rand() & 1) ? mb : -mb);
            }
            CHK(hipMemcpy(d_a, ha.data(), 1024, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_b, hb.data(), 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d);
            CHK(hipMemcpy(hd.data(), d_d, 4096, hipMemcpyDeviceToHost));
            for (int row = 0; row < 32; ++row)
                for (int col = 0; col < 32; ++col) {
                    double exact = 0.0, mag = 0.0;
                    for (int k = 0; k < 16; ++k) {
                        // lane l holds A[row = l % 32][k = 8 (l / 32) ..] and B[k = 8 (l / 32) ..][col = l % 32]
                        const double av = h2d(ha[((k >> 3) * 32 + row) * 8 + (k & 7)]), bv = h2d(hb[((k >> 3) * 32 + col) * 8 + (k & 7)]);
                        exact += av * bv; mag += fabs(av * bv);
                    }
                    const double err = fabs((double)hd[row * 32 + col] - exact);
                    if (err / mag > worst) worst = err / mag;
                    if (fabs(exact) > 0 && err / fabs(exact) > worst_rel_result && fabs(exact) > 0.25 * mag) worst_rel_result = err / fabs(exact);
                }
        }
        // binary16 subnormal inputs: honoured or flushed?  (the low parts of the filter's splits are often subnormal)
        {
            for (size_t i = 0; i < ha.size(); ++i) { ha[i] = 0; hb[i] = 0; }
            // row 0 / col 0: a_0 = 2^-20 (subnormal), b_0 = 2^10 -> product 2^-10 if honoured, 0 if flushed
            ha[(0 * 32 + 0) * 8 + 0] = f2h(ldexpf(1.f, -20)); hb[(0 * 32 + 0) * 8 + 0] = f2h(1024.f);
            // row 1 / col 1: both subnormal-free control: 2^-10 * 1
            ha[(0 * 32 + 1) * 8 + 0] = f2h(ldexpf(1.f, -10)); hb[(0 * 32 + 1) * 8 + 0] = f2h(1.f);
            // row 2 / col 2: subnormal on the B side
            ha[(0 * 32 + 2) * 8 + 0] = f2h(1024.f); hb[(0 * 32 + 2) * 8 + 0] = f2h(ldexpf(1.f, -24));
            CHK(hipMemcpy(d_a, ha.data(), 1024, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_b, hb.data(), 1024, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, d_a, d_b, d_d);
            CHK(hipMemcpy(hd.data(), d_d, 4096, hipMemcpyDeviceToHost));
            printf("binary16 subnormal inputs: A-side 2^-20 x 2^10 -> %g (honoured: %g), control 2^-10 x 1 -> %g, B-side 2^10 x 2^-24 -> %g (honoured: %g)\n",
                   hd[0 * 32 + 0], ldexp(1.0, -10), hd[1 * 32 + 1], hd[2 * 32 + 2], ldexp(1.0, -14));
        }
        printf("one MFMA vs the exact sum of its 16 products, 200 x 1024 results: max |err| / sum|a_k b_k| = %.3g = %.2f u   (u = 2^-24); "
               "max |err| / |result| where the result is not a cancellation = %.2f u\n", worst, worst / 5.9604644775390625e-08,
               worst_rel_result / 5.9604644775390625e-08);
        CHK(hipFree(d_a)); CHK(hipFree(d_b)); CHK(hipFree(d_d));
    }

    // ---- 3. VALU probes
    {
        float *d_out;
        const int blocks = cus * 8;
        CHK(hipMalloc(&d_out, sizeof(float) * 256 * blocks));
        struct { const char *name; void (*k)(float *, float, float); } probes[] = {
            { "v_fma_f32", k_fma }, { "v_or3_b32", k_or3 }, { "v_pk_fma_f16", k_pk_fma_f16 }, { "v_pk_min_f16", k_pk_min_f16 },
            { "v_dot2_f32_f16", k_dot2_f32_f16 }, { "v_cmp_lt_i32", k_cmp_lt_i32 } };
        for (auto &p : probes) {
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
            CHK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double wave_instr_per_simd = (double)ITERS * UNROLL * blocks * 4 / (cus * 4.0);
            printf("%-18s %8.3f ms   %.2f cycles per wave-instruction per SIMD (at max clock)\n", p.name, best, best * 1e-3 * clk / wave_instr_per_simd);
        }
        CHK(hipFree(d_out));
    }
    return 0;
}
