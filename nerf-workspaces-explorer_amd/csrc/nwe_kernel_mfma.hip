// Dispatch of the MFMA render kernel (templates: nwe_mfma_kernels.h).  The instantiations are compiled in separate
// translation units (nwe_mfma_inst_*.hip) so that they build in parallel.
#include "nwe_mfma_kernels.h"

namespace nwe {

#ifdef NWE_ONE_KERNEL   // register-pressure experiments: `make one` compiles just the headline instantiation to assembly
template __global__ void render_mfma_kernel<256, 8, 4, true, false, NWE_ONE_KERNEL, true>(RenderArgs, NetMfma, NetMfma);
}  // namespace nwe
#else
// Instantiated shapes: width 128 or 256, even depth 4 / 6 / 8 with the reference's skip connection (after layer 4 where
// that layer exists and feeds another trunk layer, nerf_model.py:13,58-59; none for depth 4), 63/27-wide encodings.
// The reference formulation (kFormReference) exists for the two BASELINE shapes only.
bool mfma_supported(int D, int W, int in_xyz, int in_dir, int skip, int form) {
    if (in_xyz != 63 || in_dir != (form == kFormNoViewDirs ? 0 : 27) || (W != 128 && W != 256)) return false;
    const bool shape = (D == 8 && skip == 4) || (D == 6 && skip == 4) || (D == 4 && skip == -1);
    if (!shape) return false;
    return form != kFormReference || (D == 8 && W == 256) || (D == 4 && W == 128);
}

int mfma_max_samples() { return kSplitMaxSamples; }

#define NWE_EXTERN_SHAPE(W_, D_, SKIP_, FORM_) \
    extern template bool launch_t<W_, D_, SKIP_, FORM_>(const RenderArgs&, const NetMfma&, const NetMfma&, bool, int, hipStream_t, LaunchInfo*)
NWE_EXTERN_SHAPE(256, 8, 4, kFormFolded);
#ifndef NWE_ONLY_HEADLINE
NWE_EXTERN_SHAPE(256, 8, 4, kFormReference);
NWE_EXTERN_SHAPE(256, 8, 4, kFormNoViewDirs);
NWE_EXTERN_SHAPE(256, 6, 4, kFormFolded);
NWE_EXTERN_SHAPE(256, 4, -1, kFormFolded);
NWE_EXTERN_SHAPE(128, 8, 4, kFormFolded);
NWE_EXTERN_SHAPE(128, 6, 4, kFormFolded);
NWE_EXTERN_SHAPE(128, 4, -1, kFormFolded);
NWE_EXTERN_SHAPE(128, 4, -1, kFormReference);
NWE_EXTERN_SHAPE(128, 4, -1, kFormNoViewDirs);
NWE_EXTERN_SHAPE(256, 6, 4, kFormNoViewDirs);
NWE_EXTERN_SHAPE(256, 4, -1, kFormNoViewDirs);
NWE_EXTERN_SHAPE(128, 8, 4, kFormNoViewDirs);
NWE_EXTERN_SHAPE(128, 6, 4, kFormNoViewDirs);
#endif
#undef NWE_EXTERN_SHAPE

bool launch_render_mfma(const RenderArgs& a, const NetMfma& nc, const NetMfma& nf, bool three_pass, int decomposition, hipStream_t stream,
                        LaunchInfo* info) {
    if (a.n_importance > 0 && (nf.D != nc.D || nf.W != nc.W || nf.skip != nc.skip || nf.form != nc.form)) return false;
    if (a.n_samples > kSplitMaxSamples) return false;
    const int D = nc.D, W = nc.W, skip = nc.skip;
    if (nc.form == kFormFolded) {
        if (D == 8 && W == 256 && skip == 4) return launch_t<256, 8, 4, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
#ifndef NWE_ONLY_HEADLINE
        if (D == 4 && W == 128 && skip == -1) return launch_t<128, 4, -1, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 8 && W == 128 && skip == 4) return launch_t<128, 8, 4, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 4 && W == 256 && skip == -1) return launch_t<256, 4, -1, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 6 && W == 256 && skip == 4) return launch_t<256, 6, 4, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 6 && W == 128 && skip == 4) return launch_t<128, 6, 4, kFormFolded>(a, nc, nf, three_pass, decomposition, stream, info);
#endif
        return false;
    }
#ifndef NWE_ONLY_HEADLINE
    if (nc.form == kFormNoViewDirs) {
        if (D == 8 && W == 256 && skip == 4) return launch_t<256, 8, 4, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 4 && W == 128 && skip == -1) return launch_t<128, 4, -1, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 6 && W == 256 && skip == 4) return launch_t<256, 6, 4, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 4 && W == 256 && skip == -1) return launch_t<256, 4, -1, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 8 && W == 128 && skip == 4) return launch_t<128, 8, 4, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        if (D == 6 && W == 128 && skip == 4) return launch_t<128, 6, 4, kFormNoViewDirs>(a, nc, nf, three_pass, decomposition, stream, info);
        return false;
    }
    if (D == 8 && W == 256 && skip == 4) return launch_t<256, 8, 4, kFormReference>(a, nc, nf, three_pass, decomposition, stream, info);
    if (D == 4 && W == 128 && skip == -1) return launch_t<128, 4, -1, kFormReference>(a, nc, nf, three_pass, decomposition, stream, info);
#endif
    return false;
}

}  // namespace nwe
#endif
