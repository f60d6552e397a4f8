// oa_fam_exp.hip -- the kernels of OA_FAMILY_EXP (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_all.hpp"
namespace oa {
OA_FAMILY_EXP()
}  // namespace oa
