// early_exit_probe.hip -- does a CU take a new workgroup while waves of resident ones have ALREADY ENDED?
// Workgroups of 8 waves at 80 VGPRs (6 waves per SIMD: 3 workgroups per CU by wave slots); wave 0 works for T cycles, the
// other seven have nothing to do.  A: they wait at the workgroup's barrier (slots held).  B: they end at once.  If ended
// waves give their slots back before the workgroup is gone, B runs as many workgroups per CU as the LDS allows.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/early_exit_probe.exe tools/early_exit_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <bool EXIT>
__global__ __launch_bounds__(512) void k_probe(long long spin, int *out)
{
    extern __shared__ int lds[];
    asm volatile("v_mov_b32 v79, 0" ::: "v79");                    // 80 VGPRs -> 6 waves per SIMD
    const int wave = threadIdx.x >> 6;
    if (wave == 0) {
        const long long t0 = (long long)__builtin_readcyclecounter();
        while ((long long)__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
        lds[threadIdx.x] = (int)spin;
    } else if (EXIT) return;
    if (!EXIT) __syncthreads();
    if (threadIdx.x == 0 && out) out[blockIdx.x] = lds[0];
}
int main()
{
    int *out; CHK(hipMalloc(&out, 1 << 20));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int n_wg = 256 * 24;
    for (int lds_kb : { 40, 32, 20, 8 }) {
        CHK(hipFuncSetAttribute((const void *)k_probe<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
        CHK(hipFuncSetAttribute((const void *)k_probe<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
        float ms[2] = { 0, 0 };
        for (int v = 0; v < 2; ++v) {
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(a, 0));
                if (v == 0) hipLaunchKernelGGL(k_probe<false>, dim3(n_wg), dim3(512), lds_kb << 10, 0, 20000ll, out);
                else hipLaunchKernelGGL(k_probe<true>, dim3(n_wg), dim3(512), lds_kb << 10, 0, 20000ll, out);
                CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
                CHK(hipEventElapsedTime(&ms[v], a, b));
            }
        }
        printf("LDS %2d KB per workgroup (%d per CU by LDS, 3 by wave slots): barrier %.1f us, early exit %.1f us  (%d workgroups, one wave of each busy for 20000 cycles)\n",
               lds_kb, 160 / lds_kb, ms[0] * 1e3f, ms[1] * 1e3f, n_wg);
    }
    return 0;
}
