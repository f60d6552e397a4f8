// oa_fam_exp_r8.hip -- the kernels of OA_FAMILY_EXP_R8 (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_kernels.hpp"
#include "oa_families.hpp"
namespace oa {
OA_FAMILY_EXP_R8()
}  // namespace oa
