// rtw_inst_path.hip - the instantiations of k_path (and k_classify), compiled as a translation unit of their own so that
// __graft_entry__.build() can compile the library's kernels in parallel (-DRTW_SPLIT_BUILD: rtw_hip.hip then only declares
// them). A single-file build of rtw_hip.hip (scripts/build_variant.sh, the experiments build) instantiates them itself.
#include <hip/hip_runtime.h>

#define RTW_TEMPLATES_ONLY
#include "../../include/rtw.h"
#include "rtw_device.h"
#include "rtw_kernels.h"

namespace rtwk {
#define RTW_INST(K_) \
    template __global__ void K_<RTW_RNG_PHILOX, 0>(const KArgs); template __global__ void K_<RTW_RNG_PHILOX, 1>(const KArgs); template __global__ void K_<RTW_RNG_PHILOX, 2>(const KArgs); \
    template __global__ void K_<RTW_RNG_TEA_LCG, 0>(const KArgs); template __global__ void K_<RTW_RNG_TEA_LCG, 1>(const KArgs); template __global__ void K_<RTW_RNG_TEA_LCG, 2>(const KArgs);
RTW_INST(k_path)
#undef RTW_INST
template __global__ void k_path<RTW_RNG_PHILOX, 1, 1>(const KArgs);
template __global__ void k_path<RTW_RNG_TEA_LCG, 1, 1>(const KArgs);
template __global__ void k_classify<true>(const KArgs, uint32_t*, uint32_t*, uint32_t);
template __global__ void k_classify<false>(const KArgs, uint32_t*, uint32_t*, uint32_t);
}  // namespace rtwk
