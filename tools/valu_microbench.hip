// valu_microbench.hip -- issue-rate probes for the fp32 VALU instructions k_nn_search is built from.
// Answers (on a real MI355X): is v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 twice the work per issue slot of the
// scalar forms?  What do v_min3_f32 and a broadcast ds_read_b128 cost?  Output: one line per probe with
// G lane-ops/s and the implied cycles per wave-instruction per SIMD.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_microbench.exe valu_microbench.hip     Run: ./valu_microbench.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;      // independent chains per lane

__global__ __launch_bounds__(256) void k_fma(float *out, float a, float b)
{
    float acc[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_fma(float *out, float a, float b)
{
    float2_ acc[UNROLL];
    float2_ va = { a, a * 1.5f }, vb = { b, b * 0.5f };
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = float2_{ (float)threadIdx.x + i, (float)i };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_pk_add(float *out, float a, float b)
{
    float2_ acc[UNROLL];
    float2_ va = { a, a * 1.5f };
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = float2_{ (float)threadIdx.x + i, (float)i };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(va));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + b;
}

__global__ __launch_bounds__(256) void k_pk_mul(float *out, float a, float b)
{
    float2_ acc[UNROLL];
    float2_ va = { a, a * 1.0001f };
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = float2_{ (float)threadIdx.x + i, (float)i };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(va));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s + b;
}

__global__ __launch_bounds__(256) void k_sub(float *out, float a, float b)
{
    float acc[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + b;
}

__global__ __launch_bounds__(256) void k_min3(float *out, float a, float b)
{
    float acc[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// v_sub with an SGPR operand (the "target in scalar registers" variant)
__global__ __launch_bounds__(256) void k_sub_sgpr(float *out, float a, float b)
{
    float acc[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(acc[i]) : "s"(a));
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + b;
}

// v_cmp + v_cndmask pair (the classic argmin update)
__global__ __launch_bounds__(256) void k_cmp_cnd(float *out, float a, float b)
{
    float acc[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i)
            asm volatile("v_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// broadcast ds_read_b128 (all lanes same address) + 1 dependent VALU per dword, to see the LDS issue cost
__global__ __launch_bounds__(256) void k_lds_bcast(float *out, float a, float b)
{
    __shared__ float4 tile[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = make_float4(a + i, b, a, b);
    __syncthreads();
    float acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll 16
        for (int g = 0; g < 64; ++g) {
            const float4 v = tile[(it * 64 + g) & 1023];
            acc0 += v.x; acc1 += v.y; acc2 += v.z; acc3 += v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}

template <typename F> double time_kernel(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch();
    CHK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

// integer probes: is a packed 8-bit dot product (4 MACs) issued at the v_fma_f32 rate?  (candidate for a cheaper
// first filter level: one v_dot4 + one v_min per pair instead of 2 v_fma + 1 v_min)
#define INT_PROBE(NAME, ASM)                                                                         \
__global__ __launch_bounds__(256) void NAME(float *out, float a, float b)                            \
{                                                                                                    \
    int acc[UNROLL];                                                                                 \
    const int va = __float_as_int(a) ^ threadIdx.x, vb = __float_as_int(b) + threadIdx.x;            \
    _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) acc[i] = threadIdx.x + i;                     \
    for (int it = 0; it < ITERS; ++it) {                                                             \
        _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(acc[i]) : "v"(va), "v"(vb)); \
    }                                                                                                \
    int s = 0;                                                                                       \
    _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) s += acc[i];                                  \
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;                                                  \
}
INT_PROBE(k_dot4_iu8, "v_dot4_i32_i8 %0, %1, %2, %0")
INT_PROBE(k_dot2_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
INT_PROBE(k_min_i32, "v_min_i32 %0, %0, %1")
INT_PROBE(k_min3_i32, "v_min3_i32 %0, %0, %1, %2")
INT_PROBE(k_min_f32, "v_min_f32 %0, %0, %1")
INT_PROBE(k_mad_i32_i24, "v_mad_i32_i24 %0, %1, %2, %0")

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;     // Hz
    printf("device %s  CUs %d  clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int blocks = cus * 8;                  // 8 waves per SIMD
    float *out;
    CHK(hipMalloc(&out, sizeof(float) * blocks * 256));
    struct P { const char *name; void (*k)(float *, float, float); double ops_per_inst; double insts; };
    const double n_inst = (double)ITERS * UNROLL;
    P probes[] = {
        { "v_fma_f32", k_fma, 1, n_inst }, { "v_pk_fma_f32", k_pk_fma, 2, n_inst }, { "v_sub_f32", k_sub, 1, n_inst },
        { "v_sub_f32(sgpr)", k_sub_sgpr, 1, n_inst }, { "v_pk_add_f32", k_pk_add, 2, n_inst },
        { "v_pk_mul_f32", k_pk_mul, 2, n_inst }, { "v_min3_f32", k_min3, 1, n_inst },
        { "v_cmp+v_cndmask", k_cmp_cnd, 1, 2 * n_inst },
        { "ds_read_b128 bcast+4 add", k_lds_bcast, 1, (double)(ITERS / 4) * 64 * 5 },
        { "v_dot4_i32_i8 (4 MAC)", k_dot4_iu8, 1, n_inst }, { "v_dot2_i32_i16 (2 MAC)", k_dot2_i16, 1, n_inst },
        { "v_min_i32", k_min_i32, 1, n_inst }, { "v_min3_i32", k_min3_i32, 1, n_inst }, { "v_min_f32", k_min_f32, 1, n_inst },
        { "v_mad_i32_i24", k_mad_i32_i24, 1, n_inst },
    };
    for (auto &p : probes) {
        const double ms = time_kernel([&] { hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
        const double waves = (double)blocks * 4;
        const double wave_insts = waves * p.insts;
        const double cyc_per_inst_per_simd = (ms * 1e-3 * clk) * (cus * 4) / wave_insts;
        const double glaneops = wave_insts * 64 * p.ops_per_inst / (ms * 1e-3) / 1e9;
        printf("%-26s %8.3f ms  %9.1f G lane-ops/s  %6.2f cycles per wave-instruction per SIMD (at max clock)\n",
               p.name, ms, glaneops, cyc_per_inst_per_simd);
    }
    CHK(hipFree(out));
    return 0;
}
