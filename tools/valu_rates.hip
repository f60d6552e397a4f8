// Issue rates of the vector-ALU instructions the search kernels choose between, measured on the GPU box (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/_exp/valu_rates tools/valu_rates.hip     (here: cross-compiles)
//   gpurun -- 'tools/_exp/valu_rates'                                                                   (results: profiles/r05q_valu_rates.txt)
// Each kernel runs 16 independent dependency chains of ONE instruction per lane, 8 waves per SIMD on every SIMD; the figure is
// nanoseconds per wave-instruction per SIMD (1.0 ns ~ one instruction every two cycles at ~2 GHz = the full rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int OP> __global__ __launch_bounds__(256) void burn(float *out, int trips, float seed)
{
    float a[16]; h2 p[16];
    for (int k = 0; k < 16; ++k) { a[k] = seed + k + threadIdx.x; p[k] = h2{(_Float16)(seed + k), (_Float16)(seed - k)}; }
    float b = seed * 0.5f, c = seed * 0.25f; h2 pb = h2{(_Float16)b, (_Float16)c}, pc = h2{(_Float16)c, (_Float16)b};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
                if (OP == 3) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == 4) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == 5) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 7) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(p[k]) : "v"(pb), "v"(pc));
                if (OP == 8) asm volatile("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[k]) : "v"(pb), "v"(pc));
                if (OP == 9) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == 10) asm volatile("v_pk_max_f16 %0, %0, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[k]));
                if (OP == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 12) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == 13) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[k]), "v"(b) : "vcc");
                if (OP == 14) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
                if (OP == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 16) asm volatile("v_min_f16 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 17) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 18) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(pb), "v"(pc));
                if (OP == 20) asm volatile("v_min3_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 21) asm volatile("v_sub_f16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 22) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
                if (OP == 23) asm volatile("v_minimum3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 24) asm volatile("v_min3_u16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 25) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 26) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 27) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 28) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (OP == 29) asm volatile("v_min3_f16 %0, |%0|, |%1|, |%2| op_sel:[1,1,0,0]" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 30) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                if (OP == 31) asm volatile("v_sub_u16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 32) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 33) asm volatile("v_pk_sub_u16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 34) asm volatile("v_min_u16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 35) asm volatile("v_max_f16 %0, %1, %0" : "+v"(a[k]) : "v"(b));
                if (OP == 19) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[k & ~1]) : "v"(*(double*)&b));
            }
    }
    float s = 0; for (int k = 0; k < 16; ++k) s += a[k] + (float)p[k].x + (float)p[k].y;
    if (s == 12345.678f) out[0] = s;
}
template <int OP> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int trips = 20000, blocks = 256 * 8;     // 8 waves / SIMD
    burn<OP><<<blocks, 256>>>(d, 100, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); burn<OP><<<blocks, 256>>>(d, trips, 1.0f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * trips * 64;            // wave-instructions
    const double per_simd = winst / 1024.0;
    printf("%-34s %8.3f ms  %7.2f G wave-instr/s/SIMD  (%.3f ns per instr per SIMD)\n", name, ms, per_simd / ms / 1e6, ms * 1e6 / per_simd);
    (void)hipFree(d);
}
int main()
{
    run<0>("v_add_f32 2src"); run<1>("v_fma_f32 3src"); run<17>("v_sub_f32"); run<6>("v_min_f32"); run<5>("v_min3_f32");
    run<2>("v_pk_fma_f16 3src"); run<7>("v_pk_fma_f16 (acc in src2)"); run<8>("v_pk_fma_f16 op_sel splat"); run<3>("v_pk_min_f16"); run<4>("v_pk_add_f16");
    run<12>("v_pk_mul_f16"); run<10>("v_pk_max_f16 neg (abs)"); run<9>("v_pk_min_i16"); run<14>("v_pk_min_u16"); run<11>("v_and_b32"); run<15>("v_min_u32"); run<16>("v_min_f16");
    run<13>("v_cmp_gt_f32"); run<18>("v_dot2_f32_f16");
    run<20>("v_min3_f16"); run<29>("v_min3_f16 abs op_sel"); run<21>("v_sub_f16"); run<22>("v_pk_minimum3_f16"); run<23>("v_minimum3_f32"); run<24>("v_min3_u16"); run<25>("v_min3_u32");
    run<26>("v_or3_b32"); run<27>("v_sad_u16"); run<28>("v_cvt_pkrtz_f16_f32"); run<30>("v_med3_f32"); run<31>("v_sub_u16"); run<32>("v_sub_u32"); run<33>("v_pk_sub_u16"); run<34>("v_min_u16"); run<35>("v_max_f16");
    run<0>("v_add_f32 2src (again)");
    return 0;
}
