// Host/device shared descriptors of the two render kernels and their launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nwe_device.h"

namespace nwe {

// ---- fp32 kernel: transposed weights Wt[k][n] + bias in one float blob ------------------------
struct LayerF32 {
    int K, N;
    int64_t wt_off, b_off;  // float offsets into the blob
};
struct NetF32 {
    const float* blob;
    int D, W, in_xyz, in_dir, skip;
    LayerF32 pts[16];
    LayerF32 views, feature, alpha, rgb;
    LayerF32 output;     // use_view_dirs=False (in_dir == 0): _output_linear [out_ch, W] instead of the four heads
    int out_ch;
};
constexpr int kMaxDepth = 16;

void launch_render_f32(const RenderArgs& a, const NetF32& nc, const NetF32& nf, hipStream_t stream);

// ---- MFMA kernel: a stream of 1-KiB tiles in consumption order (DESIGN.md "weight stream") -----
struct NetMfma {
    const uint8_t* stream;  // device: 1-KiB tiles, (hi, lo) per k-step, chunk after chunk
    const float* bias;      // device: 32 floats per chunk (tile row i -> bias of the weight row it holds); folded: then W/32 + 1 dot
                            // rows (the weights of _alpha_linear in the row order of the last trunk layer's tiles, and its bias)
    int n_tiles, n_chunks;
    float inv_scale;        // weights are stored multiplied by 1/inv_scale (a power of two)
    int D, W, skip;
    int form;               // Form: which formulation of the network the stream holds
};

// The three formulations the MFMA kernel is instantiated for (template argument FORM of nwe_mfma_kernels.h).
enum Form {
    kFormReference = 0,     // every layer of nerf_model.py:45-76 as a tile of the stream (selectable for comparison)
    kFormFolded = 1,        // the product path: _feature_linear multiplied into the view layer at pack time, _alpha_linear a dot product
    kFormNoViewDirs = 2     // use_view_dirs=False (nerf_model.py:41-43,78-79): trunk, then the rows rgb_raw(3), sigma_raw of _output_linear
};

// true if a kernel instantiation exists for this shape (in_dir == 0 exactly for kFormNoViewDirs)
bool mfma_supported(int D, int W, int in_xyz, int in_dir, int skip, int form);
// returns false if the shape has no instantiation.  decomposition: -1 = pick by frame size, 0 = four ray packets per
// workgroup, 1 = one packet per workgroup with its samples dealt to the four waves, 2 = full rounds as 0 and the ragged
// last round as 1 in a second launch (bit-identical results)
// *info (may be null): in, `mid` = an event to record between the two launches of plan 2 (or null); out, the plan taken
// (0 / 1 / 2 as above) and the rays the first launch got (all of them unless plan 2)
struct LaunchInfo {
    hipEvent_t mid = nullptr;
    int plan = -1;
    int64_t rays_first = 0;
    bool mid_recorded = false;
};
bool launch_render_mfma(const RenderArgs& a, const NetMfma& nc, const NetMfma& nf, bool three_pass, int decomposition, hipStream_t stream,
                        LaunchInfo* info);

constexpr int kTileBytes = 1024;
int mfma_max_samples();      // n_samples the MFMA kernel's per-wave LDS buffers are sized for

// Self-test kernels (nwe_selftest.hip)
int run_selftest(int32_t* report8, hipStream_t stream);

}  // namespace nwe
