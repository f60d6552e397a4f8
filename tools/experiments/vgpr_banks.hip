#include <hip/hip_runtime.h>
#include <cstdio>
// VGPR bank conflicts of a 3-source packed fma on gfx950: explicit physical registers (values are garbage, timing only)
#define REP16(X) X X X X X X X X X X X X X X X X
template <int V> __global__ __launch_bounds__(256) void k(float *out, int trips)
{
    for (int t = 0; t < trips; ++t) {
        if (V == 0) asm volatile(REP16("v_pk_fma_f16 v10, v16, v65, v18\n v_pk_fma_f16 v11, v17, v66, v19\n v_pk_fma_f16 v12, v20, v65, v22\n v_pk_fma_f16 v13, v21, v66, v23\n") ::: "v10","v11","v12","v13");   // no conflicts
        if (V == 1) asm volatile(REP16("v_pk_fma_f16 v10, v16, v66, v18\n v_pk_fma_f16 v11, v17, v66, v19\n v_pk_fma_f16 v12, v20, v66, v22\n v_pk_fma_f16 v13, v21, v66, v23\n") ::: "v10","v11","v12","v13");   // half: src1 = src2 bank
        if (V == 2) asm volatile(REP16("v_pk_fma_f16 v10, v16, v66, v18\n v_pk_fma_f16 v11, v16, v66, v18\n v_pk_fma_f16 v12, v20, v66, v22\n v_pk_fma_f16 v13, v20, v66, v22\n") ::: "v10","v11","v12","v13");   // all: src1 = src2 bank
        if (V == 3) asm volatile(REP16("v_pk_fma_f16 v10, v16, v64, v18\n v_pk_fma_f16 v11, v16, v64, v18\n v_pk_fma_f16 v12, v20, v64, v22\n v_pk_fma_f16 v13, v20, v64, v22\n") ::: "v10","v11","v12","v13");   // all: src0 = src1 bank
        if (V == 4) asm volatile(REP16("v_pk_fma_f16 v10, v16, v64, v20\n v_pk_fma_f16 v11, v16, v64, v20\n v_pk_fma_f16 v12, v24, v64, v28\n v_pk_fma_f16 v13, v24, v64, v28\n") ::: "v10","v11","v12","v13");   // all three in one bank
        if (V == 5) asm volatile(REP16("v_pk_minimum3_f16 v10, v16, v65, v18\n v_pk_minimum3_f16 v11, v17, v66, v19\n v_pk_minimum3_f16 v12, v20, v65, v22\n v_pk_minimum3_f16 v13, v21, v66, v23\n") ::: "v10","v11","v12","v13");
        if (V == 6) asm volatile(REP16("v_pk_minimum3_f16 v10, v16, v66, v18\n v_pk_minimum3_f16 v11, v16, v66, v18\n v_pk_minimum3_f16 v12, v20, v66, v22\n v_pk_minimum3_f16 v13, v20, v66, v22\n") ::: "v10","v11","v12","v13");
        if (V == 7) asm volatile(REP16("v_fma_f32 v10, v16, v65, v18\n v_fma_f32 v11, v17, v66, v19\n v_fma_f32 v12, v20, v65, v22\n v_fma_f32 v13, v21, v66, v23\n") ::: "v10","v11","v12","v13");
        if (V == 8) asm volatile(REP16("v_fma_f32 v10, v16, v66, v18\n v_fma_f32 v11, v16, v66, v18\n v_fma_f32 v12, v20, v66, v22\n v_fma_f32 v13, v20, v66, v22\n") ::: "v10","v11","v12","v13");
        if (V == 9) asm volatile(REP16("v_pk_fma_f16 v10, v16, v65, v18 op_sel_hi:[0,1,0]\n v_pk_fma_f16 v11, v16, v65, v18 op_sel:[1,0,1]\n v_pk_fma_f16 v12, v17, v66, v19 op_sel_hi:[0,1,0]\n v_pk_fma_f16 v13, v17, v66, v19 op_sel:[1,0,1]\n") ::: "v10","v11","v12","v13");
        if (V == 10) asm volatile(REP16("v_sub_f32 v10, v16, v66\n v_sub_f32 v11, v17, v66\n v_sub_f32 v12, v18, v66\n v_sub_f32 v13, v19, v66\n") ::: "v10","v11","v12","v13");   // one in four: same bank
        if (V == 11) asm volatile(REP16("v_sub_f32 v10, v18, v66\n v_sub_f32 v11, v18, v66\n v_sub_f32 v12, v22, v66\n v_sub_f32 v13, v22, v66\n") ::: "v10","v11","v12","v13");   // all: same bank
    }
    if (trips < 0) out[0] = 1.f;
}
template <int V> void run(const char *name)
{
    float *d; (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int trips = 5000, blocks = 256 * 8;
    k<V><<<blocks, 256>>>(d, 50); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<V><<<blocks, 256>>>(d, trips); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * trips * 64 / 1024.0;
    printf("%-64s %8.3f ms  %.3f ns per instr per SIMD\n", name, ms, ms * 1e6 / per_simd);
}
int main()
{
    run<0>("v_pk_fma_f16, three sources in three banks");
    run<1>("v_pk_fma_f16, src1/src2 share a bank in half the instructions");
    run<2>("v_pk_fma_f16, src1/src2 share a bank in all");
    run<3>("v_pk_fma_f16, src0/src1 share a bank in all");
    run<4>("v_pk_fma_f16, all three sources in one bank");
    run<9>("v_pk_fma_f16 op_sel splats, three banks");
    run<5>("v_pk_minimum3_f16, three banks");
    run<6>("v_pk_minimum3_f16, src1/src2 share a bank");
    run<7>("v_fma_f32, three banks");
    run<8>("v_fma_f32, src1/src2 share a bank");
    run<10>("v_sub_f32, sources share a bank in one of four");
    run<11>("v_sub_f32, sources share a bank in all");
    return 0;
}
