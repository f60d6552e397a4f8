// oa_fam_grid.hip -- the kernels of OA_FAMILY_GRID (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_bvh.hpp"                 // (+ oa_tri.hpp, oa_grid.hpp, oa_kernels.hpp)
#include "oa_families.hpp"
namespace oa {
OA_FAMILY_GRID()
}  // namespace oa
