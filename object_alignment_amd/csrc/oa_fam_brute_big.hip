// oa_fam_brute_big.hip -- the kernels of OA_FAMILY_BRUTE_BIG (oa_families.hpp), explicitly instantiated; nothing else lives here.
#define OA_FAMILY_TU 1
#include "oa_kernels.hpp"
#include "oa_families.hpp"
namespace oa {
OA_FAMILY_BRUTE_BIG()
}  // namespace oa
