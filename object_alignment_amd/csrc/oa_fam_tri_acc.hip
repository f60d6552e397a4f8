// oa_fam_tri_acc.hip -- the kernels of OA_FAMILY_TRI_ACC (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_bvh.hpp"                 // (+ oa_tri.hpp, oa_grid.hpp, oa_kernels.hpp)
#include "oa_families.hpp"
namespace oa {
OA_FAMILY_TRI_ACC()
}  // namespace oa
