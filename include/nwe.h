/*
 * nwe.h -- C ABI of the MI355X-native NeRF volume-rendering path (libnwe_hip.so).
 *
 * The reference (dmjovan/NeRF-Workspaces-Explorer) has no FFI layer: its hot path is plain PyTorch
 * behind the public surface of NeRFReplicaInferenceHandler.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  Plain pointers and sizes
 * only; no torch types.  All functions return 0 on success, a NWE_ERR_* code otherwise, never throw,
 * and leave a message retrievable with nwe_last_error().  A context is not thread-safe.
 */
#ifndef NWE_H
#define NWE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nwe_ctx nwe_ctx;

enum {
    NWE_OK = 0,
    NWE_ERR_INVALID = 1,     /* bad argument (null pointer, size, shape) */
    NWE_ERR_UNSUPPORTED = 2, /* network shape / precision combination has no kernel */
    NWE_ERR_HIP = 3,         /* a HIP runtime call failed; message holds hipGetErrorString */
    NWE_ERR_STATE = 4        /* call order: network / sampling tables not set */
};

enum { NWE_NET_COARSE = 0, NWE_NET_FINE = 1 };

/* Arithmetic of the MLP GEMMs.  Everything else (rays, sampling, encoding, compositing) is fp32
 * (with the fp64 running products/sums torch's CPU cumprod/cumsum use) in every mode. */
enum {
    NWE_PREC_F16X3 = 0, /* MFMA f16, operands split hi+lo, 3 products, fp32 accumulate: fp32-grade (default) */
    NWE_PREC_F16X1 = 1, /* MFMA f16 single product: ~1e-3 abs on RGB, PSNR > 50 dB */
    NWE_PREC_F32 = 2    /* fp32 FMA chain on the vector ALU, any layer shape: on-device reference */
};

/* bits OR-ed into *nwe_outputs.flags: the reference only prints
 * "[Numerical Error] <key> contains NaN or inf." (nerf_replica_inference_handler.py:273-275) */
enum {
    NWE_FLAG_RGB = 1u << 0, NWE_FLAG_DEPTH = 1u << 1, NWE_FLAG_ACC = 1u << 2, NWE_FLAG_DISP = 1u << 3,
    NWE_FLAG_RGB_COARSE = 1u << 4, NWE_FLAG_DEPTH_COARSE = 1u << 5, NWE_FLAG_ACC_COARSE = 1u << 6,
    NWE_FLAG_DISP_COARSE = 1u << 7, NWE_FLAG_RAW = 1u << 8, NWE_FLAG_ZSTD = 1u << 9
};

/* Device output buffers, owned by the caller; any pointer may be NULL (that output is skipped).
 * R = number of rays of the call, S = n_samples + n_importance.  Keys mirror the output dict of
 * NeRFReplicaInferenceHandler._volumetric_rendering (nerf_replica_inference_handler.py:256-268).
 * When n_importance == 0 the "fine" slots receive the coarse results (the reference raises
 * UnboundLocalError there, :263). */
typedef struct nwe_outputs {
    uint64_t struct_bytes; /* = sizeof(nwe_outputs): a caller built against another version of this header (fields were
                              added since) is refused with NWE_ERR_INVALID instead of having pointers misread */
    float *rgb;          /* [R,3]   rgb_fine    */
    float *depth;        /* [R]     depth_fine  */
    float *acc;          /* [R]     acc_fine    */
    float *disp;         /* [R]     disp_fine   */
    float *z_std;        /* [R]     z_std       */
    float *rgb_coarse;   /* [R,3]               */
    float *depth_coarse; /* [R]                 */
    float *acc_coarse;   /* [R]                 */
    float *disp_coarse;  /* [R]                 */
    float *raw_coarse;   /* [R,n_samples,4]  network output, [rgb_raw(3), sigma_raw] */
    float *raw_fine;     /* [R,S,4]             */
    float *z_fine;       /* [R,S]  sorted sample depths of the fine pass (handler.py:243) */
    float *weights_coarse; /* [R,n_samples]  alpha * transmittance of the coarse pass: the 4th return value of raw2outputs
                                   (nerf/models/model_utils.py:80), whose [..., 1:-1] slice is what sample_pdf takes
                                   (handler.py:237) */
    /* Conditioning diagnostics of the importance sampling (nerf/rays/rays.py:103-119), one value per ray, taken over
     * the importance samples that are interpolated between two different cdf entries (the clamped end case
     * below == above, :104-105, has a structural denom of 0 and a zero-width bin and is left out): */
    float *sample_cond;  /* [R]    smallest cdf step `denom` (:113) BEFORE the `denom < 1e-5 -> 1` replacement (:114); a value
                                   below 1e-5 says that a sample of this ray went through the replacement (1.0 if none) */
    float *sample_amp;   /* [R]    largest bin_width / denom (denom after the replacement): a sample depth moves by this
                                   much per unit change of the coarse cdf, i.e. the first-order amplification of a
                                   rounding difference in the coarse weights IN THE REFERENCE ALGORITHM ITSELF */
    float *sample_switch;/* [R]    smallest |denom - 1e-5| before the replacement: the distance of the ray's samples
                                   from the discontinuity of :114 */
    float *feat_map;     /* [R,W/2] feat_map_fine (experiment.endpoint_feat, handler.py:248-254,270-271): the view layer's outputs
                                   of the FINE pass composited like rgb (model_utils.py:87-89; the reference takes the last 128
                                   channels, i.e. W = 256).  NWE_PREC_F32 only, networks with view directions, n_importance > 0 */
    uint32_t *flags;     /* [1]    NWE_FLAG_* bits, OR-ed (caller zeroes it) */
} nwe_outputs;

/* Create a context bound to HIP device `device`.  device == -1 creates a host-only context that can
 * pack networks (for CPU tests of the packer) but cannot render.
 * Replaces: NeRFReplicaInferenceHandler.__init__ device side (handler.py:25-86). */
int nwe_create(nwe_ctx **out, int device);
void nwe_destroy(nwe_ctx *ctx);

/* Message of the last failed call on `ctx` (or of the last failed nwe_create when ctx == NULL). */
const char *nwe_last_error(const nwe_ctx *ctx);

/* Upload one NeRFModel (nerf/models/nerf_model.py:10-43, use_view_dirs=True) from HOST fp32 tensors in
 * PyTorch nn.Linear layout ([out,in] row-major, y = x W^T + b).  Copied and repacked inside; the caller
 * may free its buffers on return.  Replaces load_state_dict + .cuda() (handler.py:106-141).
 *   depth, width   D, W of the density trunk
 *   in_xyz,in_dir  encoded input widths (63, 27): 3 + 6*num_freqs
 *   skip_layer     i such that gamma(x) is concatenated in front of h AFTER layer i's ReLU
 *                  (nerf_model.py:58-59; 4 for D=8), or -1
 *   w, b           depth+4 pointers each, ordered: _pts_linears[0..depth-1], _views_linears[0],
 *                  _feature_linear, _alpha_linear, _rgb_linear */
int nwe_set_network(nwe_ctx *ctx, int which, int depth, int width, int in_xyz, int in_dir, int skip_layer,
                    const float *const *w, const float *const *b);

/* The same for NeRFModel(use_view_dirs=False) (nerf/models/nerf_model.py:41-43,82-83; the handler builds it with
 * input_ch_views = 0 and output_ch = 5, handler.py:97-119): the trunk and ONE output layer.  w, b: depth+1 pointers each,
 * _pts_linears[0..depth-1] then _output_linear [output_ch, width]; output_ch >= 4, channels 0..2 are rgb_raw and channel 3
 * sigma_raw (model_utils.py:62,71), further channels are ignored as the reference ignores them.  Such networks render with
 * every precision where the shape is one the MFMA kernel is instantiated for (widths 128 and 256, depth 6 or 8 with the skip
 * after layer 4, or depth 4 without) - it evaluates the trunk and
 * then one tile of _output_linear - and with NWE_PREC_F32 otherwise; they take 8-column rays in nwe_render_rays ([o d near far], nerf/rays/rays.py:26-30
 * without the view directions) and both networks of a context must be of the same kind. */
int nwe_set_network_no_view_dirs(nwe_ctx *ctx, int which, int depth, int width, int in_xyz, int skip_layer, int output_ch,
                                 const float *const *w, const float *const *b);

/* Sampling tables computed by the host with torch.linspace (its bits are not i/(n-1)):
 *   t_vals[n_samples] = linspace(0,1,Ns) and one_minus_t[n_samples] = 1 - t_vals  (handler.py:216-218)
 *   u[n_importance]   = linspace(0,1,Ni)                                        (nerf/rays/rays.py:95)
 * n_importance may be 0 (coarse only; u may then be NULL). */
int nwe_set_sampling(nwe_ctx *ctx, const float *t_vals, const float *one_minus_t, int n_samples,
                     const float *u, int n_importance);

/* Render rays [row_begin,row_end) x [0,W) of n_poses pinhole views.  c2w: HOST, n_poses*16 floats,
 * row-major 4x4 camera-to-world.  Output ray index = (p*(row_end-row_begin) + (h-row_begin))*W + w.
 * Ray generation follows nerf/rays/rays.py:6-71; the render loop nerf_replica_inference_handler.py:203-277.
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream) when c2w is pageable host memory (the poses are
 * staged before the call returns); a pinned c2w must stay valid until the stream has passed the call.  A context may
 * have launches in flight on several streams (each launch owns its pose table); it is still not thread-safe, and with
 * more than four launches queued the call blocks until the oldest has finished.
 * Replaces: create_rays + rays.cuda() + _render_rays of render_coordinates (handler.py:172-177). */
int nwe_render(nwe_ctx *ctx, const float *c2w, int n_poses, int H, int W, float fx, float fy, float cx, float cy,
               float near, float far, int row_begin, int row_end, int precision, const nwe_outputs *out,
               void *stream);

/* One process, several contexts: every context renders one contiguous row tile of every pose - context i the rows
 * dist.shard_rows(H, n_ctx)[i], the first H % n_ctx tiles one row longer - on a stream of its own, and copies it into the
 * caller's row-major frames on contexts[0]'s device (hipMemcpyPeerAsync: xGMI between devices, a plain device copy when the
 * contexts share a device; "several tiles on one device" and "one tile per device" are the same code).  The caller's
 * stream (on contexts[0]'s device) continues when all tiles have landed.  Every context must have its networks and
 * sampling tables set, identically.  rgb_dev [n_poses,H,W,3], depth_dev / acc_dev [n_poses,H,W], flags_dev [1]; each may be
 * NULL.  Errors are reported on contexts[0].
 * Why it exists: the reference renders from its GUI thread (application/app.py:336 -> application/workspace.py:66 ->
 * render_coordinates), which cannot be one rank of a torchrun job; this is the multi-GPU path of that call.  (One process
 * per GPU with an RCCL gather is nwe_amd/dist.py.) */
int nwe_render_tiled(nwe_ctx *const *contexts, int n_ctx, const float *c2w, int n_poses, int H, int W, float fx, float fy,
                     float cx, float cy, float near, float far, int precision, float *rgb_dev, float *depth_dev,
                     float *acc_dev, uint32_t *flags_dev, void *stream);

/* What the last nwe_render_tiled on contexts[0] went through without failing but the caller should know (peer access
 * between two devices not available or not enabled: the tile copies are then staged by the runtime); "" if nothing. */
const char *nwe_last_warning(const nwe_ctx *ctx);
/* How `tile`'s device reaches `first`'s (contexts[0]'s) memory in nwe_render_tiled: 1 direct (same device, or peer access
 * over xGMI enabled), 0 hipDeviceCanAccessPeer says no (staged copies), -1 the query or hipDeviceEnablePeerAccess failed. */
int nwe_debug_peer_access(const nwe_ctx *first, const nwe_ctx *tile);

/* Generate the rays of nwe_render() without rendering them: DEVICE rays_out [n_poses*(row_end-row_begin)*W, 11]
 * fp32 = [o(3) d(3) near far viewdir(3)], bit-identical to the reference's CPU result.
 * Replaces: create_rays (nerf/rays/rays.py:6-32). */
int nwe_create_rays(nwe_ctx *ctx, const float *c2w, int n_poses, int H, int W, float fx, float fy, float cx, float cy,
                    float near, float far, int row_begin, int row_end, float *rays_out_dev, void *stream);

/* Render precomputed rays: DEVICE [n_rays,11] fp32 = [o(3) d(3) near far viewdir(3)] (rays.py:26-30); [n_rays,8] without the
 * view directions when the context's networks were set with nwe_set_network_no_view_dirs.
 * Replaces: NeRFReplicaInferenceHandler._render_rays(flat_rays) (handler.py:187-201). */
int nwe_render_rays(nwe_ctx *ctx, const float *rays_dev, int64_t n_rays, int precision, const nwe_outputs *out,
                    void *stream);

/* uint8 = (255 * clip(x,0,1)) truncated, elementwise over n floats on the device.
 * Replaces: to8b_np (nerf/models/model_utils.py:9) of render_coordinates' tail (handler.py:183). */
int nwe_to8b(nwe_ctx *ctx, const float *rgb_dev, uint8_t *out_dev, int64_t n, void *stream);

/* Algorithmic FLOPs (2 x GEMM MACs of the reference formulation) of one MLP evaluation of network `which`. */
int64_t nwe_flops_per_eval(const nwe_ctx *ctx, int which);

/* Time of the most recent RENDER launch on this context (nwe_render / nwe_render_rays / this context's tile of
 * nwe_render_tiled; nwe_create_rays does not count), from HIP events recorded on its stream around the kernel; blocks
 * until that launch has finished.  Returns < 0 if nothing was launched.  Like every entry point it leaves the calling
 * thread's current HIP device as it found it. */
float nwe_last_kernel_ms(nwe_ctx *ctx);

/* The same launch taken apart: a frame whose last round of workgroups is ragged is rendered as TWO launches of the kernel
 * template back to back (the full rounds as four-packet workgroups, the rest sample-split; nwe_debug_last_plan == 2).
 * ms2[0], rays2[0]: the first (or only) launch; ms2[1], rays2[1]: the second, -1 / 0 if there was none.  Blocks like
 * nwe_last_kernel_ms. */
int nwe_last_launch_parts(nwe_ctx *ctx, float *ms2, int64_t *rays2);

/* --- test hooks ------------------------------------------------------------------------------- */

/* Size in bytes / host copy of the MFMA weight stream packed for network `which` (0 if that network
 * has no MFMA kernel).  Layout: see DESIGN.md "weight stream".  Works on host-only contexts. */
int64_t nwe_packed_bytes(const nwe_ctx *ctx, int which);
int nwe_packed_copy(const nwe_ctx *ctx, int which, void *host_dst, int64_t bytes);
/* The rest of the packed network: the per-chunk bias table (32 floats per chunk) and the power of two the packed
 * weights are multiplied by. */
int64_t nwe_packed_bias_count(const nwe_ctx *ctx, int which);
int nwe_packed_bias_copy(const nwe_ctx *ctx, int which, float *host_dst, int64_t count);
float nwe_packed_scale(const nwe_ctx *ctx, int which);

/* Test hook: the NEXT nwe_render_rays call takes the fine-pass sample depths from z_dev (DEVICE [n_rays, S],
 * sorted per ray) instead of its own importance sampling; cleared after that call.  Lets a test feed the
 * reference's own depths and compare the fine pass alone. */
int nwe_debug_set_fine_depths(nwe_ctx *ctx, const float *z_dev);

/* Test hooks in the same style (the NEXT nwe_render_rays call, cleared after it); both kernels honour them:
 *   nwe_debug_set_raw             network outputs from the caller instead of evaluating the MLP: DEVICE raw_coarse
 *                                 [n_rays, n_samples, 4] and / or raw_fine [n_rays, S, 4] (either may be NULL), so that
 *                                 compositing (nerf/models/model_utils.py:49-100) is checked on the reference's own
 *                                 edge vectors (sigma <= 0 everywhere, saturated alpha, sigma_last = +-1e-11);
 *   nwe_debug_set_coarse_weights  coarse weights [n_rays, n_samples] from the caller instead of running the coarse pass
 *                                 (its outputs are then not written), so that the inverse-CDF sampling
 *                                 (nerf/rays/rays.py:74-121) is checked on given weights. */
int nwe_debug_set_raw(nwe_ctx *ctx, const float *raw_coarse_dev, const float *raw_fine_dev);
int nwe_debug_set_coarse_weights(nwe_ctx *ctx, const float *weights_dev);

/* Test hook: networks uploaded AFTER this call are packed for the MFMA kernel with (1, the default) or without (0)
 * _feature_linear multiplied into _views_linears (nerf/models/nerf_model.py:64-70: the feature layer has no activation,
 * so W_v [W_f h + b_f; gamma(d)] + b_v = (W_v[:, :W] W_f) h + ...; fp64 product on the host).  0 evaluates the feature
 * layer as the reference formulates it (8x256 and 4x128 only); kept for one-to-one comparison. */
int nwe_debug_set_fold(nwe_ctx *ctx, int on);

/* rendering.white_background of the reference's YAML (nerf/models/model_utils.py:97-98): when on, every rgb output
 * (coarse and fine) is rgb + (1 - acc).  Off by default, as in all four office configs. */
int nwe_set_white_background(nwe_ctx *ctx, int on);

/* Training-mode forward (nerf/training/nerf_replica_training_handler.py:553-580; forward only): the NEXT
 * nwe_render_rays call uses random numbers drawn by the caller exactly where the reference calls torch.rand /
 * torch.randn, one row per ray of that call (DEVICE pointers, each may be NULL = the inference behaviour); cleared
 * after that call.
 *   t_rand       [n_rays, n_samples]                stratified jitter in [0,1): z = lower + (upper - lower) * t_rand (:553-562)
 *   noise_coarse [n_rays, n_samples]                added to sigma_raw before the ReLU, i.e. randn * raw_noise_std
 *   noise_fine   [n_rays, n_samples + n_importance] (nerf/models/model_utils.py:64-71)
 *   u_sorted     [n_rays, n_importance]             the uniform numbers of sample_pdf(det=False) (nerf/rays/rays.py:98),
 *                                                   sorted ascending per ray: sample_pdf is element-wise in u and the
 *                                                   reference sorts the union of depths afterwards (:580), so the order of
 *                                                   u does not change any output */
int nwe_set_train_tables(nwe_ctx *ctx, const float *t_rand_dev, const float *noise_coarse_dev, const float *noise_fine_dev,
                         const float *u_sorted_dev);

/* Test hook: the MFMA kernel has two work decompositions with bit-identical results (four ray packets per workgroup, or
 * one packet whose samples are dealt to the four waves) and picks by frame size - all of one kind, or the full rounds of
 * workgroups as packets and the ragged last round sample-split in a second launch; mode 0 / 1 / 2 forces a plan, -1
 * restores the automatic choice. */
int nwe_debug_set_decomposition(nwe_ctx *ctx, int mode);
/* The plan the most recent MFMA launch of this context took (0 packets / 1 sample split / 2 packets + split rest), -1 if
 * none yet: lets a test check the launcher's choice without timing anything. */
int nwe_debug_last_plan(const nwe_ctx *ctx);

/* Diagnostic builds only (make -C csrc stamps): DEVICE buffer of 10 uint64 per wave that a -DNWE_STAMPS build of the MFMA
 * kernel fills with s_memtime cycle sums (tools/stamp_run.py); the product build never touches it.  NULL switches it off. */
int nwe_debug_set_stamps(nwe_ctx *ctx, unsigned long long *per_wave_dev);

/* Device self-test of the hardware assumptions the MFMA kernel relies on (fragment layouts of
 * v_mfma_f32_32x32x16_f16, fp16 subnormal operands, LDS-DMA lane order).  report[0..7] receives
 * mismatch counts / measured values; returns NWE_OK when every assumption holds. */
int nwe_selftest(nwe_ctx *ctx, int32_t *report8);

#ifdef __cplusplus
}
#endif
#endif /* NWE_H */
