// rtw_inst_shade.hip - the instantiations of the wavefront kernels that shade (k_first, k_shade, k_bounce), compiled as a
// translation unit of their own (see rtw_inst_path.hip).
#include <hip/hip_runtime.h>

#define RTW_TEMPLATES_ONLY
#include "../../include/rtw.h"
#include "rtw_device.h"
#include "rtw_kernels.h"

namespace rtwk {
// -DRTW_INST_KIND=0 / 1: only the Philox / only the TEA+LCG instantiations (__graft_entry__.build() compiles this file twice, side
// by side: it is the longest of the translation units); without the macro, both
#if !defined(RTW_INST_KIND) || RTW_INST_KIND == 0
#define RTW_INST(K_) template __global__ void K_<RTW_RNG_PHILOX, 0>(const KArgs); template __global__ void K_<RTW_RNG_PHILOX, 1>(const KArgs); template __global__ void K_<RTW_RNG_PHILOX, 2>(const KArgs);
RTW_INST(k_first)
RTW_INST(k_shade)
RTW_INST(k_bounce)
#undef RTW_INST
#endif
#if !defined(RTW_INST_KIND) || RTW_INST_KIND == 1
#define RTW_INST(K_) template __global__ void K_<RTW_RNG_TEA_LCG, 0>(const KArgs); template __global__ void K_<RTW_RNG_TEA_LCG, 1>(const KArgs); template __global__ void K_<RTW_RNG_TEA_LCG, 2>(const KArgs);
RTW_INST(k_first)
RTW_INST(k_shade)
RTW_INST(k_bounce)
#undef RTW_INST
#endif
}  // namespace rtwk
