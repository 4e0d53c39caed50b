// Instantiations of the MFMA render kernel, part b (see nwe_kernel_mfma.hip).
#include "nwe_mfma_kernels.h"

namespace nwe {
template bool launch_t<256, 8, 4, kFormReference>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
template bool launch_t<256, 4, -1, kFormFolded>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
}  // namespace nwe
