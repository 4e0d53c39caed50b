// Instantiations of the MFMA render kernel, part f (see nwe_kernel_mfma.hip): networks without view directions, the other shapes.
#include "nwe_mfma_kernels.h"

namespace nwe {
template bool launch_t<256, 6, 4, kFormNoViewDirs>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
template bool launch_t<256, 4, -1, kFormNoViewDirs>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
template bool launch_t<128, 8, 4, kFormNoViewDirs>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*);
}  // namespace nwe
