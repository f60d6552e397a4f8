// oa_all.hpp -- every kernel header, in one order, for all translation units of liboa_icp.so (oa_families.hpp says which
// unit compiles what).  oa_sort.hpp stays with the host unit: its launches sit in an inline host function.
#pragma once
#include "oa_kernels.hpp"
#include "oa_grid.hpp"
#include "oa_tri.hpp"
#include "oa_tri_ring.hpp"
#include "oa_tri_fine.hpp"
#include "oa_bvh.hpp"
#include "oa_affine.hpp"
#include "oa_mfma.hpp"
#include "oa_families.hpp"
