// sort_bench.hip -- rocprim::radix_sort_pairs on 30-bit Morton keys: the library's default algorithm choice (merge sort up to
// 1Mi items) against Onesweep forced through radix_sort_config<.., MergeSortLimit>.  Same (stable) result, bit for bit.
// Third contender: the library's own three-pass LSD sort (object_alignment_amd/csrc/oa_sort.hpp), which must return the same
// permutation (both are stable) -- keys AND values are compared.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/sort_bench.exe tools/sort_bench.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include "../object_alignment_amd/csrc/oa_sort.hpp"
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using cfg_onesweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 32768>;
template <class Config>
int run(const char *tag, unsigned *k_in, unsigned *k_out, int *v_in, int *v_out, size_t n, std::vector<int> &res)
{
    size_t bytes = 0;
    CHK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, 30, 0)));
    void *tmp; CHK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        CHK(hipEventRecord(a, 0));
        CHK((rocprim::radix_sort_pairs<Config>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0, 30, 0)));
        CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    res.resize(n); CHK(hipMemcpy(res.data(), v_out, n * sizeof(int), hipMemcpyDeviceToHost));
    printf("  %-28s n = %8zu  %7.1f us  (temporary storage %zu B)\n", tag, n, best * 1e3f, bytes);
    CHK(hipFree(tmp));
    return 0;
}
int run_lsd(unsigned *k_in, unsigned *k_out, int *v_in, int *v_out, size_t n, std::vector<int> &res, std::vector<unsigned> &resk)
{
    const size_t bytes = oa::sort_order_tmp_bytes(n);
    void *tmp; CHK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        CHK(hipEventRecord(a, 0));
        CHK(oa::sort_order_lsd(tmp, k_in, v_out, n, 30, 0));
        CHK(hipEventRecord(b, 0)); CHK(hipEventSynchronize(b));
        float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    res.resize(n); CHK(hipMemcpy(res.data(), v_out, n * sizeof(int), hipMemcpyDeviceToHost));
    resk.resize(n);
    printf("  %-28s n = %8zu  %7.1f us  (temporary storage %zu B)\n", "oa_sort.hpp (3 x 10 bits)", n, best * 1e3f, bytes);
    CHK(hipFree(tmp));
    return 0;
}
int main()
{
    for (size_t n : { (size_t)1, (size_t)63, (size_t)2049, (size_t)2562, (size_t)20000, (size_t)41000, (size_t)100000, (size_t)300000, (size_t)1000000, (size_t)1957200, (size_t)10000000 }) {
        std::mt19937 rng(7);
        std::vector<unsigned> hk(n); std::vector<int> hv(n);
        for (size_t i = 0; i < n; ++i) { hk[i] = rng() & 0x3FFFFFFFu; if (i % 5 == 0) hk[i] = hk[i / 2]; hv[i] = (int)i; }   // with ties
        unsigned *k_in, *k_out; int *v_in, *v_out;
        CHK(hipMalloc(&k_in, n * 4)); CHK(hipMalloc(&k_out, n * 4)); CHK(hipMalloc(&v_in, n * 4)); CHK(hipMalloc(&v_out, n * 4));
        CHK(hipMemcpy(k_in, hk.data(), n * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(v_in, hv.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<int> r0, r1;
        if (run<rocprim::default_config>("default (merge <= 1Mi)", k_in, k_out, v_in, v_out, n, r0)) return 1;
        if (run<cfg_onesweep>("onesweep above 32k", k_in, k_out, v_in, v_out, n, r1)) return 1;
        std::vector<int> r2; std::vector<unsigned> k2;
        if (run_lsd(k_in, k_out, v_in, v_out, n, r2, k2)) return 1;
        bool keys_ok = true;
        for (size_t i = 0; i < n; ++i) { k2[i] = hk[(size_t)r2[i]]; if (i && k2[i - 1] > k2[i]) { keys_ok = false; break; } }
        std::vector<unsigned> hk_after(n); CHK(hipMemcpy(hk_after.data(), k_in, n * 4, hipMemcpyDeviceToHost));
        printf("  same permutation: rocprim merge / onesweep %s, oa_sort %s; keys ascending in that order %s; inputs untouched %s\n", r0 == r1 ? "yes" : "NO",
               r0 == r2 ? "yes" : "NO", keys_ok ? "yes" : "NO", hk_after == hk ? "yes" : "NO");
        hipFree(k_in); hipFree(k_out); hipFree(v_in); hipFree(v_out);
    }
    return 0;
}
