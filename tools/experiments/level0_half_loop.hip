#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// the level-0 block of k_nn_search_sorted16, alone: VARIANT 0 tile from LDS, 1 tile "from registers" (opaque to the optimiser),
// 2 = LDS reads only (no math), 3 = from LDS with two independent min chains per point pair
template <int VARIANT, int WG_PER_CU> __global__ __launch_bounds__(256, WG_PER_CU) void k(float *out, int trips, const float4 *src)
{
    __shared__ float4 tile[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) tile[i] = src[i];
    __syncthreads();
    half2v Y[2], T[2];
    Y[0] = half2v{(_Float16)(threadIdx.x * 0.01f), (_Float16)(threadIdx.x * 0.02f)}; Y[1] = half2v{(_Float16)(threadIdx.x * 0.03f), (_Float16)1.f};
    T[0] = half2v{(_Float16)-1000.f, (_Float16)-1000.f}; T[1] = T[0];
    int hits = 0;
    for (int t = 0; t < trips; ++t) {
        for (int g = 0; g < 256; g += 4) {
            float4 P[4];
            for (int k = 0; k < 4; ++k) P[k] = tile[g + k];
            if (VARIANT == 1) for (int k = 0; k < 4; ++k) { P[k] = tile[k]; asm volatile("" : "+v"(P[k].x), "+v"(P[k].y), "+v"(P[k].z), "+v"(P[k].w)); }
            if (VARIANT == 2) { asm volatile("" :: "v"(P[0].x), "v"(P[1].y), "v"(P[2].z), "v"(P[3].w)); continue; }
            bool hit0 = false;
            for (int p = 0; p < 2; ++p) {
                half2v s[16];
                for (int k = 0; k < 4; ++k) {
                    const half2v C01 = __builtin_bit_cast(half2v, P[k].x), C23 = __builtin_bit_cast(half2v, P[k].y);
                    const half2v W01 = __builtin_bit_cast(half2v, P[k].z), W23 = __builtin_bit_cast(half2v, P[k].w);
                    s[4 * k] = __builtin_elementwise_fma(half2v{C01.x, C01.x}, Y[p], half2v{W01.x, W01.x});
                    s[4 * k + 1] = __builtin_elementwise_fma(half2v{C01.y, C01.y}, Y[p], half2v{W01.y, W01.y});
                    s[4 * k + 2] = __builtin_elementwise_fma(half2v{C23.x, C23.x}, Y[p], half2v{W23.x, W23.x});
                    s[4 * k + 3] = __builtin_elementwise_fma(half2v{C23.y, C23.y}, Y[p], half2v{W23.y, W23.y});
                }
                half2v m;
                if (VARIANT == 3) {
                    half2v m1 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(s[0], s[1]), s[2]);
                    half2v m2 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(s[3], s[4]), s[5]);
                    m1 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m1, s[6]), s[7]);
                    m2 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m2, s[8]), s[9]);
                    m1 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m1, s[10]), s[11]);
                    m2 = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m2, s[12]), s[13]);
                    m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m1, s[14]), s[15]);
                    m = __builtin_elementwise_minimum(m, m2);
                } else {
                    m = s[0];
                    for (int k = 1; k + 1 < 16; k += 2) m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(m, s[k]), s[k + 1]);
                    m = __builtin_elementwise_minimum(m, s[15]);
                }
                hit0 |= !(m.x > T[p].x) | !(m.y > T[p].y);
            }
            if (hit0) { hits++; Y[0].x += (_Float16)1.f; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = hits;
}
template <int VARIANT, int W> void run(const char *name)
{
    float *d; float4 *s; (void)hipMalloc(&d, 4 * 256 * 4096); (void)hipMalloc(&s, 16 * 1024); (void)hipMemset(s, 0x3c, 16 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int trips = 200, blocks = 256 * W;
    k<VARIANT, W><<<blocks, 256>>>(d, 2, s); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<VARIANT, W><<<blocks, 256>>>(d, trips, s); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // pairs per SIMD (wave-level): waves per SIMD = W; per wave: trips * 64 blocks * 64 pairs
    const double wave_pairs_per_simd = (double)W * trips * 64 * 64;
    printf("%-50s waves/SIMD %d  %8.3f ms  %.3f ns per pair per SIMD\n", name, W, ms, ms * 1e6 / wave_pairs_per_simd);
    (void)hipFree(d); (void)hipFree(s);
}
int main()
{
    run<0, 4>("tile from LDS"); run<1, 4>("tile in registers"); run<2, 4>("LDS reads only"); run<3, 4>("tile from LDS, two min chains");
    run<0, 2>("tile from LDS"); run<1, 2>("tile in registers"); run<3, 2>("tile from LDS, two min chains");
    run<0, 1>("tile from LDS"); run<1, 1>("tile in registers");
    return 0;
}
