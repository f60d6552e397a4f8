// oa_families.hpp -- which translation unit compiles which kernel.
//
// liboa_icp.so is built from one host translation unit (oa_icp.hip: context, uploads, launch geometry, the C-ABI, and
// every PLAIN kernel -- those are guarded by !OA_FAMILY_TU in the headers) and one translation unit per family of
// heavy kernel TEMPLATES (oa_fam_*.hip), compiled in parallel by __graft_entry__.build_hip / the Makefile.  A family
// unit explicitly instantiates its list below; the host unit declares the same list `extern`, so it launches the
// kernels (hipLaunchKernelGGL on a declared, not defined, instantiation: the host stub and the kernel handle resolve at
// link time, no relocatable device code needed) without compiling them.  One line per instantiation, right here: the
// launch sites in oa_icp.hip and this list must agree, and the linker says so when they do not.
//
// OA_EXPERIMENTS (the second flavour, liboa_icp_exp.so): A/B predecessors and measured-but-not-kept variants -- the
// matrix-core filter (oa_mfma.hpp), the seed + neighbour lists (oa_tri_ring.hpp), the settled-pose front search over a sparse
// fine grid (oa_tri_fine.hpp, round 6), k_nn_search_filtered, the instrumented (STATS) and unshared (SHARE = false) searches.  The default library neither compiles nor dispatches them.
#pragma once

// ---- signatures (explicit instantiation needs the parameter types) --------------------------------------------------
#define OA_SIG_NN_SEARCH const DevState *, const float4 *, const float4 *, int, unsigned long long *
#define OA_SIG_NN_FILTERED const DevState *, const float4 *, const float4 *, const float4 *, const float4 *, const float4 *, int, unsigned long long *
#define OA_SIG_WAVE_ORDER const DevState *, const float4 *, int, unsigned short *
#define OA_SIG_SEED_SORTED const DevState *, const float4 *, int, const float4 *, const float4 *, const int4 *, int, int, unsigned long long *
#define OA_SIG_NN_SORTED const DevState *, const float4 *, const float4 *, const float4 *, const float4 *, const int4 *, const float4 *, int, int, int, unsigned long long *, int, const unsigned short *, int, int, const int *, int *
#define OA_SIG_NN_GRID const DevState *, const float4 *, int, GridParams, const int *, const float4 *, float4 *, unsigned long long *, int *, int *, int, BvhParams, const float4 *, const float4 *, NormalTest, double *, unsigned long long *, const float *, uint2 *
#define OA_SIG_TRI_GRID const DevState *, const float4 *, int, GridParams, const int *, const float4 *, const float4 *, int *, unsigned long long *, int *, int *, int, unsigned long long *, BvhParams, const float4 *, const float4 *, NormalTest, double *, const int *, const int *, const int *, int, int, int
#define OA_SIG_TRI_SETTLE const DevState *, const float4 *, int, FineParams, const uint4 *, const float4 *, const float4 *, const int *, unsigned long long *, int *, int, int *, int *, unsigned long long *
#define OA_SIG_TRI_RING_BUILD float4 *, int, GridParams, const int *, const float4 *, double, int *, unsigned long long *
#define OA_SIG_BVH_SEARCH const DevState *, const float4 *, int, BvhParams, const float4 *, const float4 *, const float4 *, int *, float4 *, unsigned long long *, const int *, const int *, int, NormalTest, double *, const float *, uint2 *
#define OA_SIG_AFFINE_SOLVE const double *, const double *, int, long long, int, int, double *, double *
#define OA_SIG_NN_MFMA const DevState *, const float4 *, const float4 *, const half8 *, const float4 *, int, double, unsigned long long *

// X = `extern` (host unit) or empty (the family's own unit)
#define OA_K(X, name, sig, ...) X template __global__ void name<__VA_ARGS__>(sig);

// ---- brute force over vertices: oa_fam_brute.hip (the north-star kernel and what surrounds it) ----------------------
#define OA_FAMILY_BRUTE(X)                                                                                              \
    OA_K(X, k_nn_search, OA_SIG_NN_SEARCH, 1) OA_K(X, k_nn_search, OA_SIG_NN_SEARCH, 2)                                 \
    OA_K(X, k_nn_search, OA_SIG_NN_SEARCH, 4) OA_K(X, k_nn_search, OA_SIG_NN_SEARCH, 8)                                 \
    OA_K(X, k_sorted_wave_order, OA_SIG_WAVE_ORDER, 2) OA_K(X, k_sorted_wave_order, OA_SIG_WAVE_ORDER, 4)               \
    OA_K(X, k_sorted_wave_order, OA_SIG_WAVE_ORDER, 8)                                                                  \
    OA_K(X, k_nn_seed_sorted, OA_SIG_SEED_SORTED, 64) OA_K(X, k_nn_seed_sorted, OA_SIG_SEED_SORTED, FTILE_GROUPS)       \
    OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 1, 64) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 2, 64)
// (the unrolled search for 4 points per thread is the longest compile of the default library: a unit each)
#define OA_FAMILY_BRUTE_B(X) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 4, 64)
#define OA_FAMILY_BRUTE_BIG(X)                                                                                          \
    OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 1, FTILE_GROUPS) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 2, FTILE_GROUPS)
#define OA_FAMILY_BRUTE_BIG_B(X) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 4, FTILE_GROUPS)
// 8 points per thread (OA_NN_R=8: measured slower, half the waves per SIMD; 55 s of compile per instantiation): experiments only
#define OA_FAMILY_EXP_R8(X) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 8, 64)
#define OA_FAMILY_EXP_R8_BIG(X) OA_K(X, k_nn_search_sorted, OA_SIG_NN_SORTED, 8, FTILE_GROUPS)

// ---- uniform grid over vertices: oa_fam_grid.hip --------------------------------------------------------------------
#define OA_FAMILY_GRID(X)                                                                                               \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 1, false) OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 2, false)             \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 4, false)                                                                 \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 1, true, 256) OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 2, true, 256)     \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 4, true, 256)                                                             \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 1, true, 512) OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 2, true, 512)     \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 4, true, 512)

// ---- closest point on triangles: oa_fam_tri.hip ---------------------------------------------------------------------
#define OA_FAMILY_TRI(X)                                                                                                \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1) OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 2)                       \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 4)                                                                      \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1, false, true, false, 64) OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 2, false, true, false, 64) \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 4, false, true, false, 64)
#define OA_FAMILY_TRI_ACC(X)                                                                                            \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1, false, true, true) OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 2, false, true, true) \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 4, false, true, true)

// ---- 64-ary box trees: oa_fam_bvh.hip -------------------------------------------------------------------------------
#define OA_FAMILY_BVH(X)                                                                                                \
    OA_K(X, k_bvh_search, OA_SIG_BVH_SEARCH, false, false) OA_K(X, k_bvh_search, OA_SIG_BVH_SEARCH, false, true)        \
    OA_K(X, k_bvh_search, OA_SIG_BVH_SEARCH, true, false) OA_K(X, k_bvh_search, OA_SIG_BVH_SEARCH, true, true)

// ---- affine_matrix_from_points beyond the loop's 3-D solve: oa_fam_affine.hip ---------------------------------------
#define OA_FAMILY_AFFINE(X)                                                                                             \
    OA_K(X, k_affine_solve, OA_SIG_AFFINE_SOLVE, AFF_MAXD, false) OA_K(X, k_affine_solve, OA_SIG_AFFINE_SOLVE, 16, true) \
    OA_K(X, k_affine_solve, OA_SIG_AFFINE_SOLVE, 32, true) OA_K(X, k_affine_solve, OA_SIG_AFFINE_SOLVE, 64, true)

// ---- experiments (OA_EXPERIMENTS only): oa_fam_exp.hip --------------------------------------------------------------
#define OA_FAMILY_EXP(X)                                                                                                \
    OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 1, 64) OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 2, 64)   \
    OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 4, 64) OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 8, 64)   \
    OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 1, FTILE_GROUPS) OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 2, FTILE_GROUPS) \
    OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 4, FTILE_GROUPS) OA_K(X, k_nn_search_filtered, OA_SIG_NN_FILTERED, 8, FTILE_GROUPS) \
    OA_K(X, k_nn_search_mfma, OA_SIG_NN_MFMA, 2) OA_K(X, k_nn_search_mfma, OA_SIG_NN_MFMA, 3)                           \
    OA_K(X, k_nn_search_mfma, OA_SIG_NN_MFMA, 4)                                                                        \
    OA_K(X, k_nn_search_grid, OA_SIG_NN_GRID, 1, true, 512, true)                                                       \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1, true, true) OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1, true, false) \
    OA_K(X, k_tri_search_grid, OA_SIG_TRI_GRID, 1, false, false)                                                        \
    OA_K(X, k_tri_ring_build, OA_SIG_TRI_RING_BUILD, false) OA_K(X, k_tri_ring_build, OA_SIG_TRI_RING_BUILD, true)     \
    OA_K(X, k_tri_settle, OA_SIG_TRI_SETTLE, false) OA_K(X, k_tri_settle, OA_SIG_TRI_SETTLE, true)

#if !defined(OA_FAMILY_TU)
namespace oa {
OA_FAMILY_BRUTE(extern)
OA_FAMILY_BRUTE_B(extern)
OA_FAMILY_BRUTE_BIG(extern)
OA_FAMILY_BRUTE_BIG_B(extern)
OA_FAMILY_GRID(extern)
OA_FAMILY_TRI(extern)
OA_FAMILY_TRI_ACC(extern)
OA_FAMILY_BVH(extern)
OA_FAMILY_AFFINE(extern)
#if defined(OA_EXPERIMENTS)
OA_FAMILY_EXP(extern)
OA_FAMILY_EXP_R8(extern)
OA_FAMILY_EXP_R8_BIG(extern)
#endif
}  // namespace oa
#endif
