// oa_fam_affine.hip -- the kernels of OA_FAMILY_AFFINE (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_affine.hpp"
#include "oa_families.hpp"
namespace oa {
OA_FAMILY_AFFINE()
}  // namespace oa
