// Instantiations of the MFMA render kernel, part d (see nwe_kernel_mfma.hip).
#include "nwe_mfma_kernels.h"

namespace nwe {
template bool launch_t<128, 8, 4, kFormFolded>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
template bool launch_t<128, 6, 4, kFormFolded>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
template bool launch_t<128, 4, -1, kFormFolded>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
}  // namespace nwe
