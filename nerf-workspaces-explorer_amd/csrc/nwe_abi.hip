// C ABI of libnwe_hip.so (include/nwe.h): context, weight packing, table upload, kernel launches.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "nwe_host.h"

using namespace nwe;

namespace {

struct NetState {
    bool set = false;
    int D = 0, W = 0, in_xyz = 0, in_dir = 0, skip = -1;
    int out_ch = 0;                // use_view_dirs=False (in_dir == 0): rows of _output_linear
    int64_t flops = 0;
    std::vector<float> blob;       // fp32 kernel: per layer Wt[k][n] then bias
    NetF32 f32 = {};
    float* d_blob = nullptr;
    bool mfma_ok = false;
    int form = kFormFolded;        // MFMA stream: which formulation it holds (nwe_host.h: Form)
    std::vector<uint8_t> stream;   // MFMA kernel: 1-KiB tiles in consumption order
    std::vector<float> bias_tab;   // MFMA kernel: 32 floats per chunk, then (folded) the dot rows of _alpha_linear
    int n_chunks = 0;              // chunks of the stream = bias rows in front of the dot rows
    float w_scale = 1.f;           // power of two the packed weights are multiplied by
    NetMfma mf = {};
    uint8_t* d_stream = nullptr;
    float* d_bias = nullptr;
};

thread_local std::string g_create_error;

}  // namespace

struct nwe_ctx {
    int device = -1;
    bool host_only = false;
    NetState net[2];
    float *d_t = nullptr, *d_omt = nullptr, *d_u = nullptr;
    int ns = 0, ni = 0;
    // One slot per launch in flight: the pose table a kernel reads and the events around it.  A slot is reused only
    // after its own launch has finished (its `done` event), so renders of one context queued on different streams never
    // share a pose buffer.
    struct Slot {
        float* d_poses = nullptr;
        int poses_cap = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;   // ev_mid: between the two launches of the hybrid plan
        bool used = false;
        bool has_mid = false;
        int64_t rays_first = 0, rays_total = 0;
    };
    static constexpr int kSlots = 4;
    Slot slots[kSlots];
    int next_slot = 0, last_slot = -1;   // last_slot: the most recent RENDER launch (nwe_last_kernel_ms), not nwe_create_rays
    const float* dbg_z_fine = nullptr;
    const float *dbg_raw_c = nullptr, *dbg_raw_f = nullptr, *dbg_w = nullptr;   // nwe_debug_set_raw / _coarse_weights, one call
    int fold = 1;             // nwe_debug_set_fold: read by nwe_set_network
    // nwe_render_tiled: this context's row tile (rgb | depth | acc slabs + its flag word), its own stream, and the event
    // that says "tile rendered and copied into the frame"
    float* tile_buf = nullptr;
    size_t tile_cap = 0;       // floats
    uint32_t* tile_flags = nullptr;
    hipStream_t tile_stream = nullptr;
    hipEvent_t tile_done = nullptr;
    hipEvent_t frame_ready = nullptr;   // contexts[0] only: recorded on the caller's stream when the call starts
    uint32_t* flag_parts = nullptr;     // contexts[0] only: one flag word per tile, on its device
    int flag_parts_cap = 0;
    int peer_access = -2;               // this context's device -> contexts[0]'s device: 1 direct, 0 staged, -1 query/enable failed, -2 not asked yet
    std::string warn;                   // nwe_last_warning: what did not fail the call but the caller should know
    int white_bkgd = 0;
    int decomposition = -1;   // nwe_debug_set_decomposition
    int last_plan = -1;       // nwe_debug_last_plan
    unsigned long long* stamps = nullptr;   // nwe_debug_set_stamps
    const float *trn_t = nullptr, *trn_nc = nullptr, *trn_nf = nullptr, *trn_u = nullptr;   // nwe_set_train_tables, one call
    std::string err;
};

namespace {

int fail(nwe_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

// Every entry point that switches the calling thread's HIP device puts it back on return: a GUI or bench thread that
// queries a tile of another device must not find its later torch calls on that device.
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, NWE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// packing for the MFMA kernel.  Must mirror nwe_mfma_kernels.h (encode(), split_tile(), tile_mma()).
// ---------------------------------------------------------------------------------------------

// Column of gamma(v) (embedding.py:24-48 order: identity(3), then per band sin(3), cos(3)) that lane half h
// holds in element j of k-step s.  nb = bands per lane half (5 for xyz, 2 for dirs).  -1 = padding.
int gamma_col(int nb, int s, int h, int j) {
    const int q = s * 8 + j;
    if (q < 6 * nb) {
        const int pair = q >> 1, bl = pair / 3, c = pair % 3, band = bl + nb * h;
        return 3 + 6 * band + ((q & 1) ? 3 : 0) + c;
    }
    if (q == 6 * nb) return h ? 2 : 0;
    if (q == 6 * nb + 1) return h ? -1 : 1;
    return -1;
}

// Feature index that element j of k-step s holds in lane half h when a 32x32 accumulator tile is reused
// as the next B operand: tile rt = s/2, register r = 8*(s&1) + j, row = (r&3) + 8*(r>>2) + 4*h.
int hidden_col(int s, int h, int j) { return 16 * s + 8 * (j >> 2) + 4 * h + (j & 3); }

struct Segment {
    int kind;     // 0 = hidden, 1 = gamma(x), 2 = gamma(d)
    int ksteps;
    int col_off;  // column offset of this segment in the layer's [out,in] weight
};

struct RowMap {   // which weight row feeds tile row i (or -1)
    int n_out;
    int dup4;     // 1: rows 4..7 repeat rows 0..3 (head tiles read by both lane halves)
    int operator()(int rt, int i) const {
        int r = rt * 32 + i;
        if (dup4) { if (i >= 8) return -1; r = i & 3; }
        return r < n_out ? r : -1;
    }
};

// T = float (a layer as the caller handed it over) or double (a product of two layers, see pack_mfma): the value times
// the power-of-two scale is exact in T, hi = fp16(v), lo = fp16(v - hi) with v - hi exact in T.
template <class T>
void put_tile_pair(std::vector<uint8_t>& out, const T* w, int ld, const RowMap& rows, int rt, const Segment& sg, int s, float scale) {
    const size_t base = out.size();
    out.resize(base + 2 * kTileBytes, 0);
    _Float16* hi = reinterpret_cast<_Float16*>(out.data() + base);
    _Float16* lo = reinterpret_cast<_Float16*>(out.data() + base + kTileBytes);
    for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, h = lane >> 5;
        const int row = rows(rt, i);
        for (int j = 0; j < 8; ++j) {
            int col = sg.kind == 0 ? hidden_col(s, h, j) : gamma_col(sg.kind == 1 ? 5 : 2, s, h, j);
            T v = 0;
            if (row >= 0 && col >= 0) v = w[(size_t)row * ld + sg.col_off + col] * (T)scale;   // power of two: exact
            const _Float16 vh = (_Float16)v;
            hi[lane * 8 + j] = vh;
            lo[lane * 8 + j] = (_Float16)(v - (T)vh);
        }
    }
}

template <class T>
void put_chunk(NetState& n, const T* w, const T* b, int ld, const RowMap& rows, int rt, const std::vector<Segment>& segs) {
    for (int i = 0; i < 32; ++i) { const int r = rows(rt, i); n.bias_tab.push_back(r >= 0 ? (float)b[r] : 0.f); }
    for (const Segment& sg : segs)
        for (int s = 0; s < sg.ksteps; ++s) put_tile_pair(n.stream, w, ld, rows, rt, sg, s, n.w_scale);
}

// Stream order = the order mlp_eval() consumes chunks in.
//
// kFormFolded: _feature_linear has no activation (nerf/models/nerf_model.py:64) and its output feeds only the view layer
// (:66-70), so  W_v [W_f h + b_f ; gamma(d)] + b_v = (W_v[:, :W] W_f) h + W_v[:, W:] gamma(d) + (b_v + W_v[:, :W] b_f):
// the product is formed here in fp64 and split into (hi, lo) directly from the double, the feature layer's chunks
// disappear from the stream (8 of 78 chunks, 11 % of the MFMAs of an 8x256 evaluation).
//
// kFormNoViewDirs (use_view_dirs=False): layer D is _output_linear [out_ch, W]; its rows 0..3 (rgb_raw, sigma_raw) form the one
// chunk behind the trunk, duplicated into tile rows 4..7 for the upper lane half; further channels are ignored as the
// reference ignores them (model_utils.py:62,71).
void pack_mfma(NetState& n, const float* const* w, const float* const* b) {
    const int D = n.D, W = n.W, KH = W / 16;
    const int iv = D, ife = D + 1, ia = D + 2, irgb = D + 3;
    const bool folded = n.form == kFormFolded, noview = n.form == kFormNoViewDirs;
    n.stream.clear();
    n.bias_tab.clear();
    std::vector<double> wv, bv;   // folded view layer [W/2, W + in_dir] and its bias
    if (folded) {
        const int ldv = W + n.in_dir;
        wv.assign((size_t)(W / 2) * ldv, 0.0);
        bv.assign(W / 2, 0.0);
        for (int r = 0; r < W / 2; ++r) {
            const float* vr = w[iv] + (size_t)r * ldv;
            double* o = wv.data() + (size_t)r * ldv;
            double acc_b = (double)b[iv][r];
            for (int k = 0; k < W; ++k) {
                const double vk = (double)vr[k];
                const float* fr = w[ife] + (size_t)k * W;
                for (int c = 0; c < W; ++c) o[c] += vk * (double)fr[c];
                acc_b += vk * (double)b[ife][k];
            }
            for (int c = 0; c < n.in_dir; ++c) o[W + c] = (double)vr[W + c];
            bv[r] = acc_b;
        }
    }
    // One power-of-two scale for the whole network: the largest that keeps every scaled weight below 2^14, so that
    // the lo halves (|lo| <= ulp(hi)/2) are fp16-normal for all but vanishing weights.  The kernel multiplies the
    // accumulator by 1/scale before adding the bias; both scalings are exact.
    const int in_dims[4] = {W + n.in_dir, W, W, W / 2}, out_dims[4] = {W / 2, W, 1, 3};
    double wmax = 0.0;
    for (int li = 0; li < (noview ? D + 1 : D + 4); ++li) {
        if (folded && (li == iv || li == ife || li == ia)) continue;   // folded: multiplied out / evaluated in fp32 (dot rows)
        const size_t cnt = li < D ? (size_t)W * (li == 0 ? n.in_xyz : (li == n.skip + 1 ? W + n.in_xyz : W))
                                  : (noview ? (size_t)4 * W : (size_t)in_dims[li - D] * out_dims[li - D]);
        for (size_t k = 0; k < cnt; ++k) wmax = std::max(wmax, (double)std::fabs(w[li][k]));
    }
    for (double v : wv) wmax = std::max(wmax, std::fabs(v));
    int e = 0;
    if (wmax > 0.0 && std::isfinite(wmax)) { e = 14 - (int)std::ceil(std::log2(wmax)); e = std::min(std::max(e, -14), 30); }
    n.w_scale = std::ldexp(1.f, e);
    auto layer = [&](int li, int n_out, int ld, int n_tiles, int dup4, const std::vector<Segment>& segs) {
        RowMap rows{n_out, dup4};
        for (int rt = 0; rt < n_tiles; ++rt) put_chunk(n, w[li], b[li], ld, rows, rt, segs);
    };
    layer(0, W, n.in_xyz, W / 32, 0, {{1, 4, 0}});
    for (int i = 1; i < D; ++i) {
        if (i == n.skip + 1) layer(i, W, W + n.in_xyz, W / 32, 0, {{1, 4, 0}, {0, KH, n.in_xyz}});   // cat([pts, h]), nerf_model.py:59
        else layer(i, W, W, W / 32, 0, {{0, KH, 0}});
    }
    if (noview) {
        layer(D, 4, W, 1, 1, {{0, KH, 0}});
        n.n_chunks = (int)(n.bias_tab.size() / 32);
        return;
    }
    if (!folded) {
        layer(ife, W, W, W / 32, 0, {{0, KH, 0}});
        layer(ia, 1, W, 1, 1, {{0, KH, 0}});
    }
    if (folded) {
        RowMap rows{W / 2, 0};
        for (int rt = 0; rt < W / 64; ++rt) put_chunk(n, wv.data(), bv.data(), W + n.in_dir, rows, rt, {{0, KH, 0}, {2, 2, W}});
    } else {
        layer(iv, W / 2, W + n.in_dir, W / 64, 0, {{0, KH, 0}, {2, 2, W}});                           // cat([feature, views]), :66
    }
    layer(irgb, 3, W / 2, 1, 1, {{0, KH / 2, 0}});
    n.n_chunks = (int)(n.bias_tab.size() / 32);
    if (folded) {
        // _alpha_linear (nerf_model.py:63) is not a tile of the folded stream: the kernel accumulates sigma = w . h + b in fp32
        // with the epilogues of the last trunk layer's tiles.  Row rt of the dot table = the weights of trunk features
        // 32 rt .. 32 rt + 31 (the row order of that layer's tile rt, like its bias row), then one row with the bias in front.
        for (int k = 0; k < W; ++k) n.bias_tab.push_back(w[ia][k]);
        n.bias_tab.push_back(b[ia][0]);
        n.bias_tab.resize(n.bias_tab.size() + 31, 0.f);
    }
}

void pack_f32(NetState& n, const float* const* w, const float* const* b) {
    n.blob.clear();
    auto add = [&](int li, int K, int N) {
        LayerF32 L; L.K = K; L.N = N; L.wt_off = (int64_t)n.blob.size();
        n.blob.resize(n.blob.size() + (size_t)K * N);
        float* wt = n.blob.data() + L.wt_off;
        for (int k = 0; k < K; ++k) for (int o = 0; o < N; ++o) wt[(size_t)k * N + o] = w[li][(size_t)o * K + k];
        L.b_off = (int64_t)n.blob.size();
        n.blob.insert(n.blob.end(), b[li], b[li] + N);
        while (n.blob.size() % 4) n.blob.push_back(0.f);
        return L;
    };
    const int D = n.D, W = n.W;
    n.f32 = {};
    n.f32.D = D; n.f32.W = W; n.f32.in_xyz = n.in_xyz; n.f32.in_dir = n.in_dir; n.f32.skip = n.skip;
    n.f32.out_ch = n.out_ch;
    n.f32.pts[0] = add(0, n.in_xyz, W);
    for (int i = 1; i < D; ++i) n.f32.pts[i] = add(i, i == n.skip + 1 ? W + n.in_xyz : W, W);
    if (n.in_dir == 0) {                    // nerf_model.py:82-83: outputs = _output_linear(h)
        n.f32.output = add(D, W, n.out_ch);
        return;
    }
    n.f32.views = add(D, W + n.in_dir, W / 2);
    n.f32.feature = add(D + 1, W, W);
    n.f32.alpha = add(D + 2, W, 1);
    n.f32.rgb = add(D + 3, W / 2, 3);
}

int64_t algo_flops(const NetState& n) {   // 2 x MACs of nerf_model.py:53-76
    int64_t mac = (int64_t)n.in_xyz * n.W;
    for (int i = 1; i < n.D; ++i) mac += (int64_t)(i == n.skip + 1 ? n.W + n.in_xyz : n.W) * n.W;
    if (n.in_dir == 0) return 2 * (mac + (int64_t)n.W * n.out_ch);
    mac += n.W /*alpha*/ + (int64_t)n.W * n.W /*feature*/ + (int64_t)(n.W + n.in_dir) * (n.W / 2) + (int64_t)(n.W / 2) * 3;
    return 2 * mac;
}

__global__ void to8b_kernel(const float* __restrict__ x, uint8_t* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {   // model_utils.py:9: (255 * clip(x, 0, 1)).astype(uint8) -- truncation
        const float v = 255.f * fminf(fmaxf(x[i], 0.f), 1.f);
        y[i] = (uint8_t)(int)v;
    }
}

__global__ void or_flags_kernel(const uint32_t* __restrict__ parts, int n, uint32_t* __restrict__ out) {
    uint32_t f = 0;
    for (int i = 0; i < n; ++i) f |= parts[i];
    if (f) atomicOr(out, f);
}

__global__ void create_rays_kernel(RenderArgs a, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_rays) return;
    const Ray r = load_ray(a, i);
    float* o = out + i * 11;
    o[0] = r.ox; o[1] = r.oy; o[2] = r.oz; o[3] = r.dx; o[4] = r.dy; o[5] = r.dz;
    o[6] = r.near; o[7] = r.far; o[8] = r.vx; o[9] = r.vy; o[10] = r.vz;
}

// The slot of the next launch: waits for the launch that used it kSlots launches ago (normally long finished).
int acquire_slot(nwe_ctx* c, nwe_ctx::Slot** out, bool render = true) {
    nwe_ctx::Slot& s = c->slots[c->next_slot];
    if (!s.ev0) {
        HIPCHK(c, hipEventCreate(&s.ev0));
        HIPCHK(c, hipEventCreate(&s.ev1));
        HIPCHK(c, hipEventCreate(&s.ev_mid));
    }
    if (s.used) HIPCHK(c, hipEventSynchronize(s.ev1));
    if (render) c->last_slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % nwe_ctx::kSlots;
    *out = &s;
    return NWE_OK;
}

// Poses into the slot's own table.  hipMemcpyAsync from pageable host memory is staged by the runtime before it returns,
// so the caller's array is free on return; from pinned memory the copy is truly asynchronous and include/nwe.h asks the
// caller to keep the array alive until the stream has passed it.
int upload_poses(nwe_ctx* c, nwe_ctx::Slot& s, const float* c2w, int n_poses, hipStream_t stream) {
    if (n_poses > s.poses_cap) {
        if (s.d_poses) { (void)hipFree(s.d_poses); s.d_poses = nullptr; s.poses_cap = 0; }   // its last reader finished (acquire_slot)
        const int cap = std::max(n_poses, 64);
        HIPCHK(c, hipMalloc(&s.d_poses, (size_t)cap * 16 * sizeof(float)));
        s.poses_cap = cap;
    }
    HIPCHK(c, hipMemcpyAsync(s.d_poses, c2w, (size_t)n_poses * 16 * sizeof(float), hipMemcpyHostToDevice, stream));
    return NWE_OK;
}

int check_ready(nwe_ctx* ctx, const nwe_outputs* out, int precision) {
    if (!ctx || !out) return fail(ctx, NWE_ERR_INVALID, "null context or outputs");
    if (out->struct_bytes != sizeof(nwe_outputs))
        return fail(ctx, NWE_ERR_INVALID, "nwe_outputs.struct_bytes != sizeof(nwe_outputs): the caller was built against another version of include/nwe.h");
    if (ctx->host_only) return fail(ctx, NWE_ERR_STATE, "host-only context cannot render");
    if (ctx->ns <= 0) return fail(ctx, NWE_ERR_STATE, "nwe_set_sampling has not been called");
    if (!ctx->net[0].set) return fail(ctx, NWE_ERR_STATE, "coarse network not set");
    if (ctx->ni > 0 && !ctx->net[1].set) return fail(ctx, NWE_ERR_STATE, "fine network not set but n_importance > 0");
    if (precision != NWE_PREC_F16X3 && precision != NWE_PREC_F16X1 && precision != NWE_PREC_F32)
        return fail(ctx, NWE_ERR_INVALID, "unknown precision");
    if (ctx->ni > 0 && (ctx->net[0].in_dir == 0) != (ctx->net[1].in_dir == 0))
        return fail(ctx, NWE_ERR_STATE, "coarse and fine networks must both have, or both lack, view directions");
    if (out->feat_map) {
        if (ctx->ni <= 0 || ctx->net[1].in_dir == 0)
            return fail(ctx, NWE_ERR_INVALID, "feat_map is the fine pass's view-layer output: it needs n_importance > 0 and networks with view directions");
        if (precision != NWE_PREC_F32)
            return fail(ctx, NWE_ERR_UNSUPPORTED, "feat_map (endpoint_feat) is computed by the NWE_PREC_F32 kernel only");
    }
    if (precision != NWE_PREC_F32) {
        if (!ctx->net[0].mfma_ok || (ctx->ni > 0 && !ctx->net[1].mfma_ok))
            return fail(ctx, NWE_ERR_UNSUPPORTED,
                        "no MFMA kernel for this network shape (have widths 128 and 256 with depth 6 or 8 and the skip after layer 4, or depth 4 without, 63 + 27 or, without view directions, 63 inputs); use NWE_PREC_F32");
        if (ctx->ns > mfma_max_samples())
            return fail(ctx, NWE_ERR_UNSUPPORTED, "the MFMA kernel supports n_samples <= 128; use NWE_PREC_F32");
    }
    return NWE_OK;
}

int launch(nwe_ctx* ctx, nwe_ctx::Slot& slot, RenderArgs& a, int precision, void* stream_) {
    a.white_bkgd = ctx->white_bkgd;
    hipStream_t stream = (hipStream_t)stream_;
    a.t_vals = ctx->d_t; a.omt_vals = ctx->d_omt; a.u_vals = ctx->d_u;
    a.n_samples = ctx->ns; a.n_importance = ctx->ni;
    a.stamps = ctx->stamps;   // only read by -DNWE_STAMPS builds of the kernel (nwe_debug_set_stamps)
    if (a.n_rays <= 0) return NWE_OK;
    HIPCHK(ctx, hipEventRecord(slot.ev0, stream));
    slot.has_mid = false; slot.rays_first = slot.rays_total = a.n_rays;
    if (precision == NWE_PREC_F32) {
        launch_render_f32(a, ctx->net[0].f32, ctx->net[ctx->ni > 0 ? 1 : 0].f32, stream);
    } else {
        LaunchInfo info;
        info.mid = slot.ev_mid;
        if (!launch_render_mfma(a, ctx->net[0].mf, ctx->net[ctx->ni > 0 ? 1 : 0].mf, precision == NWE_PREC_F16X3, ctx->decomposition, stream,
                                &info))
            return fail(ctx, NWE_ERR_UNSUPPORTED, "coarse and fine networks must have the same shape for the MFMA kernel");
        ctx->last_plan = info.plan;
        slot.has_mid = info.mid_recorded; slot.rays_first = info.rays_first;
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(slot.ev1, stream));
    slot.used = true;
    return NWE_OK;
}

}  // namespace

extern "C" {

int nwe_create(nwe_ctx** out, int device) {
    if (!out) return fail(nullptr, NWE_ERR_INVALID, "out is null");
    nwe_ctx* c = new nwe_ctx();
    c->device = device;
    c->host_only = device < 0;
    if (!c->host_only) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            g_create_error = std::string("nwe_create: ") + hipGetErrorString(e);
            delete c;
            return NWE_ERR_HIP;
        }
    }
    *out = c;
    return NWE_OK;
}

void nwe_destroy(nwe_ctx* c) {
    if (!c) return;
    if (!c->host_only) {
        DeviceGuard guard;
        (void)hipSetDevice(c->device);
        for (NetState& n : c->net) { if (n.d_blob) (void)hipFree(n.d_blob); if (n.d_stream) (void)hipFree(n.d_stream); if (n.d_bias) (void)hipFree(n.d_bias); }
        if (c->d_t) (void)hipFree(c->d_t);
        if (c->tile_stream) (void)hipStreamSynchronize(c->tile_stream);
        if (c->tile_buf) (void)hipFree(c->tile_buf);
        if (c->tile_flags) (void)hipFree(c->tile_flags);
        if (c->flag_parts) (void)hipFree(c->flag_parts);
        if (c->tile_done) (void)hipEventDestroy(c->tile_done);
        if (c->frame_ready) (void)hipEventDestroy(c->frame_ready);
        if (c->tile_stream) (void)hipStreamDestroy(c->tile_stream);
        for (nwe_ctx::Slot& sl : c->slots) {
            if (sl.used) (void)hipEventSynchronize(sl.ev1);
            if (sl.d_poses) (void)hipFree(sl.d_poses);
            if (sl.ev0) (void)hipEventDestroy(sl.ev0);
            if (sl.ev1) (void)hipEventDestroy(sl.ev1);
            if (sl.ev_mid) (void)hipEventDestroy(sl.ev_mid);
        }
    }
    delete c;
}

const char* nwe_last_error(const nwe_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static int set_network_impl(nwe_ctx* c, int which, int depth, int width, int in_xyz, int in_dir, int skip_layer, int out_ch,
                            const float* const* w, const float* const* b) {
    const int n_layers = in_dir == 0 ? depth + 1 : depth + 4;
    for (int i = 0; i < n_layers; ++i)
        if (!w[i] || !b[i]) return fail(c, NWE_ERR_INVALID, "null weight or bias pointer");
    NetState& n = c->net[which];
    n.set = false;   // stays false if anything below fails
    n.D = depth; n.W = width; n.in_xyz = in_xyz; n.in_dir = in_dir; n.skip = skip_layer; n.out_ch = out_ch;
    n.flops = algo_flops(n);
    pack_f32(n, w, b);
    n.form = in_dir == 0 ? kFormNoViewDirs : (c->fold != 0 ? kFormFolded : kFormReference);
    n.mfma_ok = mfma_supported(depth, width, in_xyz, in_dir, skip_layer, n.form);
    if (n.mfma_ok) pack_mfma(n, w, b); else { n.stream.clear(); n.bias_tab.clear(); n.n_chunks = 0; }
    n.mf = {};
    n.mf.D = depth; n.mf.W = width; n.mf.skip = skip_layer; n.mf.form = n.form;
    n.mf.n_tiles = (int)(n.stream.size() / kTileBytes);
    n.mf.n_chunks = n.mfma_ok ? n.n_chunks : 0;
    n.mf.inv_scale = 1.f / n.w_scale;
    if (!c->host_only) {
        DeviceGuard guard;
        HIPCHK(c, hipSetDevice(c->device));
        if (n.d_blob) { (void)hipFree(n.d_blob); n.d_blob = nullptr; }
        if (n.d_stream) { (void)hipFree(n.d_stream); n.d_stream = nullptr; }
        if (n.d_bias) { (void)hipFree(n.d_bias); n.d_bias = nullptr; }
        HIPCHK(c, hipMalloc(&n.d_blob, n.blob.size() * sizeof(float)));
        HIPCHK(c, hipMemcpy(n.d_blob, n.blob.data(), n.blob.size() * sizeof(float), hipMemcpyHostToDevice));
        if (n.mfma_ok) {
            HIPCHK(c, hipMalloc(&n.d_stream, n.stream.size()));
            HIPCHK(c, hipMemcpy(n.d_stream, n.stream.data(), n.stream.size(), hipMemcpyHostToDevice));
            HIPCHK(c, hipMalloc(&n.d_bias, n.bias_tab.size() * sizeof(float)));
            HIPCHK(c, hipMemcpy(n.d_bias, n.bias_tab.data(), n.bias_tab.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        n.f32.blob = n.d_blob;
        n.mf.stream = n.d_stream;
        n.mf.bias = n.d_bias;
    }
    n.set = true;
    return NWE_OK;
}

int nwe_set_network(nwe_ctx* c, int which, int depth, int width, int in_xyz, int in_dir, int skip_layer,
                    const float* const* w, const float* const* b) {
    if (!c || !w || !b) return fail(c, NWE_ERR_INVALID, "null argument");
    if (which != NWE_NET_COARSE && which != NWE_NET_FINE) return fail(c, NWE_ERR_INVALID, "which must be 0 or 1");
    if (depth < 1 || depth > kMaxDepth) return fail(c, NWE_ERR_UNSUPPORTED, "depth must be in 1..16");
    if (width < 2 || width > 256 || width % 2) return fail(c, NWE_ERR_UNSUPPORTED, "width must be even and <= 256");
    if (in_xyz < 3 || in_xyz > 93 || (in_xyz - 3) % 6 || in_dir < 3 || in_dir > 63 || (in_dir - 3) % 6)
        return fail(c, NWE_ERR_UNSUPPORTED, "encoded widths must be 3 + 6*num_freqs (xyz <= 93, dir <= 63)");
    if (skip_layer < -1 || skip_layer >= depth - 1) skip_layer = -1;   // a skip after the last trunk layer never feeds a layer
    return set_network_impl(c, which, depth, width, in_xyz, in_dir, skip_layer, 0, w, b);
}

int nwe_set_network_no_view_dirs(nwe_ctx* c, int which, int depth, int width, int in_xyz, int skip_layer, int output_ch,
                                 const float* const* w, const float* const* b) {
    if (!c || !w || !b) return fail(c, NWE_ERR_INVALID, "null argument");
    if (which != NWE_NET_COARSE && which != NWE_NET_FINE) return fail(c, NWE_ERR_INVALID, "which must be 0 or 1");
    if (depth < 1 || depth > kMaxDepth) return fail(c, NWE_ERR_UNSUPPORTED, "depth must be in 1..16");
    if (width < 2 || width > 256 || width % 2) return fail(c, NWE_ERR_UNSUPPORTED, "width must be even and <= 256");
    if (in_xyz < 3 || in_xyz > 93 || (in_xyz - 3) % 6) return fail(c, NWE_ERR_UNSUPPORTED, "encoded width must be 3 + 6*num_freqs (<= 93)");
    if (output_ch < 4 || output_ch > 256) return fail(c, NWE_ERR_UNSUPPORTED, "output_ch must be in 4..256 (rgb_raw, sigma_raw, ignored rest)");
    if (skip_layer < -1 || skip_layer >= depth - 1) skip_layer = -1;
    return set_network_impl(c, which, depth, width, in_xyz, 0, skip_layer, output_ch, w, b);
}

int nwe_set_sampling(nwe_ctx* c, const float* t_vals, const float* one_minus_t, int n_samples, const float* u,
                     int n_importance) {
    if (!c || !t_vals || !one_minus_t) return fail(c, NWE_ERR_INVALID, "null argument");
    if (n_samples < 2 || n_samples > kMaxSamples) return fail(c, NWE_ERR_UNSUPPORTED, "n_samples must be in 2..128");
    if (n_importance < 0 || n_importance > kMaxImportance) return fail(c, NWE_ERR_UNSUPPORTED, "n_importance must be in 0..256");
    if (n_importance > 0 && (!u || n_samples < 3)) return fail(c, NWE_ERR_INVALID, "importance sampling needs u and n_samples >= 3");
    c->ns = n_samples; c->ni = n_importance;
    if (c->host_only) return NWE_OK;
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_t) {
        HIPCHK(c, hipMalloc(&c->d_t, (2 * kMaxSamples + kMaxImportance) * sizeof(float)));
        c->d_omt = c->d_t + kMaxSamples;
        c->d_u = c->d_omt + kMaxSamples;
    }
    HIPCHK(c, hipMemcpy(c->d_t, t_vals, n_samples * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_omt, one_minus_t, n_samples * sizeof(float), hipMemcpyHostToDevice));
    if (n_importance > 0) HIPCHK(c, hipMemcpy(c->d_u, u, n_importance * sizeof(float), hipMemcpyHostToDevice));
    return NWE_OK;
}

int nwe_render(nwe_ctx* c, const float* c2w, int n_poses, int H, int W, float fx, float fy, float cx, float cy, float near,
               float far, int row_begin, int row_end, int precision, const nwe_outputs* out, void* stream) {
    int rc = check_ready(c, out, precision);
    if (rc) return rc;
    if (!c2w || n_poses < 1 || H < 1 || W < 1 || row_begin < 0 || row_end > H || row_begin > row_end)
        return fail(c, NWE_ERR_INVALID, "bad pose / image / row range");
    if (!(fx != 0.f) || !(fy != 0.f)) return fail(c, NWE_ERR_INVALID, "fx and fy must be non-zero");
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    nwe_ctx::Slot* slot = nullptr;
    rc = acquire_slot(c, &slot);
    if (rc) return rc;
    rc = upload_poses(c, *slot, c2w, n_poses, (hipStream_t)stream);
    if (rc) return rc;
    RenderArgs a = {};
    a.rays = nullptr; a.poses = slot->d_poses;
    a.H = H; a.W = W; a.row_begin = row_begin; a.rows = row_end - row_begin;
    a.n_rays = (int64_t)n_poses * a.rows * W;
    if (a.n_rays > INT32_MAX) return fail(c, NWE_ERR_INVALID, "more than 2^31 - 1 rays in one call");
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.near = near; a.far = far;
    a.out = *out;
    return launch(c, *slot, a, precision, stream);
}

int nwe_render_tiled(nwe_ctx* const* ctxs, int n_ctx, const float* c2w, int n_poses, int H, int W, float fx, float fy, float cx,
                     float cy, float near, float far, int precision, float* rgb_dev, float* depth_dev, float* acc_dev,
                     uint32_t* flags_dev, void* stream_) {
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return fail(nullptr, NWE_ERR_INVALID, "nwe_render_tiled: no contexts");
    nwe_ctx* c0 = ctxs[0];
    for (int i = 0; i < n_ctx; ++i)
        if (!ctxs[i] || ctxs[i]->host_only) return fail(c0, NWE_ERR_INVALID, "nwe_render_tiled: null or host-only context");
    if (!c2w || n_poses < 1 || H < 1 || W < 1) return fail(c0, NWE_ERR_INVALID, "bad pose / image");
    hipStream_t stream0 = (hipStream_t)stream_;
    DeviceGuard guard;
    c0->warn.clear();
    // the caller's buffers may still be in use by earlier work on its stream: every tile stream starts behind this point
    HIPCHK(c0, hipSetDevice(c0->device));
    if (!c0->frame_ready) HIPCHK(c0, hipEventCreateWithFlags(&c0->frame_ready, hipEventDisableTiming));
    if (c0->flag_parts_cap < n_ctx) {
        if (c0->flag_parts) { (void)hipFree(c0->flag_parts); c0->flag_parts = nullptr; c0->flag_parts_cap = 0; }
        HIPCHK(c0, hipMalloc(&c0->flag_parts, (size_t)n_ctx * sizeof(uint32_t)));
        c0->flag_parts_cap = n_ctx;
    }
    HIPCHK(c0, hipMemsetAsync(c0->flag_parts, 0, (size_t)n_ctx * sizeof(uint32_t), stream0));
    HIPCHK(c0, hipEventRecord(c0->frame_ready, stream0));
    const int base = H / n_ctx, extra = H % n_ctx;      // dist.shard_rows: the first H % n tiles get one more row
    int r0 = 0, rc_all = NWE_OK;
    for (int i = 0; i < n_ctx && rc_all == NWE_OK; ++i) {
        nwe_ctx* c = ctxs[i];
        const int rows = base + (i < extra ? 1 : 0), r1 = r0 + rows;
        const size_t px = (size_t)rows * W;             // pixels of one pose's tile
        auto step = [&]() -> int {
            HIPCHK(c, hipSetDevice(c->device));
            if (!c->tile_stream) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->tile_stream, hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->tile_done, hipEventDisableTiming));
                HIPCHK(c, hipMalloc(&c->tile_flags, sizeof(uint32_t)));
            }
            if (c->peer_access == -2) {
                // direct xGMI copies into the frame; without peer access the runtime stages the copy through the host.  Neither
                // outcome fails the call, but the caller can read it (nwe_debug_peer_access, nwe_last_warning).
                c->peer_access = 1;
                if (c->device != c0->device) {
                    int can = 0;
                    hipError_t e = hipDeviceCanAccessPeer(&can, c->device, c0->device);
                    if (e == hipSuccess && can) {
                        e = hipDeviceEnablePeerAccess(c0->device, 0);
                        if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
                    }
                    if (e != hipSuccess) {
                        (void)hipGetLastError();
                        c->peer_access = -1;
                        c0->warn += "tile " + std::to_string(i) + ": peer access device " + std::to_string(c->device) + " -> " + std::to_string(c0->device) +
                                    " not enabled (" + hipGetErrorString(e) + "), copies are staged; ";
                    } else if (!can) {
                        c->peer_access = 0;
                        c0->warn += "tile " + std::to_string(i) + ": device " + std::to_string(c->device) + " cannot access device " + std::to_string(c0->device) +
                                    " directly, copies are staged; ";
                    }
                }
            }
            // Whatever happens below, stream0 must wait for everything this call has queued on the tile stream: the event is
            // recorded on EVERY exit of this step once the stream exists (a failing copy after earlier copies were queued
            // would otherwise leave stream0 waiting on the previous frame's already-completed event).
            struct RecordOnExit {
                nwe_ctx* c;
                ~RecordOnExit() { if (c->tile_done && c->tile_stream && hipEventRecord(c->tile_done, c->tile_stream) != hipSuccess) (void)hipGetLastError(); }
            } record_on_exit{c};
            const size_t need = (size_t)n_poses * px * 5;
            if (c->tile_cap < need) {
                HIPCHK(c, hipStreamSynchronize(c->tile_stream));
                if (c->tile_buf) { (void)hipFree(c->tile_buf); c->tile_buf = nullptr; c->tile_cap = 0; }
                HIPCHK(c, hipMalloc(&c->tile_buf, need * sizeof(float)));
                c->tile_cap = need;
            }
            HIPCHK(c, hipStreamWaitEvent(c->tile_stream, c0->frame_ready, 0));
            if (rows > 0) {
                HIPCHK(c, hipMemsetAsync(c->tile_flags, 0, sizeof(uint32_t), c->tile_stream));
                nwe_outputs o = {};
                o.struct_bytes = sizeof(nwe_outputs);
                float* t_rgb = c->tile_buf;
                float* t_depth = t_rgb + (size_t)n_poses * px * 3;
                float* t_acc = t_depth + (size_t)n_poses * px;
                o.rgb = rgb_dev ? t_rgb : nullptr; o.depth = depth_dev ? t_depth : nullptr; o.acc = acc_dev ? t_acc : nullptr;
                o.flags = c->tile_flags;
                const int rc = nwe_render(c, c2w, n_poses, H, W, fx, fy, cx, cy, near, far, r0, r1, precision, &o, c->tile_stream);
                if (rc) return rc;
                // tile -> frame: pose p's rows [r0, r1) are contiguous in the row-major [n_poses, H, W, C] frame
                for (int p = 0; p < n_poses; ++p) {
                    const size_t dst_px = ((size_t)p * H + r0) * W, src_px = (size_t)p * px;
                    if (rgb_dev)
                        HIPCHK(c, hipMemcpyPeerAsync(rgb_dev + dst_px * 3, c0->device, t_rgb + src_px * 3, c->device, px * 3 * sizeof(float), c->tile_stream));
                    if (depth_dev)
                        HIPCHK(c, hipMemcpyPeerAsync(depth_dev + dst_px, c0->device, t_depth + src_px, c->device, px * sizeof(float), c->tile_stream));
                    if (acc_dev)
                        HIPCHK(c, hipMemcpyPeerAsync(acc_dev + dst_px, c0->device, t_acc + src_px, c->device, px * sizeof(float), c->tile_stream));
                }
                HIPCHK(c, hipMemcpyPeerAsync(c0->flag_parts + i, c0->device, c->tile_flags, c->device, sizeof(uint32_t), c->tile_stream));
            }
            return NWE_OK;   // record_on_exit records tile_done
        };
        rc_all = step();
        if (rc_all != NWE_OK && c != c0) c0->err = "tile " + std::to_string(i) + ": " + c->err;
        r0 = r1;
    }
    // the caller's stream continues when every tile has landed (also on the error path: nothing may still be writing;
    // every tile that queued anything has re-recorded its event, see record_on_exit)
    (void)hipSetDevice(c0->device);
    for (int i = 0; i < n_ctx; ++i)
        if (ctxs[i]->tile_done) (void)hipStreamWaitEvent(stream0, ctxs[i]->tile_done, 0);
    if (rc_all != NWE_OK) return rc_all;
    if (flags_dev) {
        hipLaunchKernelGGL(or_flags_kernel, dim3(1), dim3(1), 0, stream0, c0->flag_parts, n_ctx, flags_dev);
        HIPCHK(c0, hipGetLastError());
    }
    return NWE_OK;
}

int nwe_create_rays(nwe_ctx* c, const float* c2w, int n_poses, int H, int W, float fx, float fy, float cx, float cy, float near,
                    float far, int row_begin, int row_end, float* rays_out_dev, void* stream) {
    if (!c || c->host_only) return fail(c, NWE_ERR_STATE, "needs a device context");
    if (!c2w || !rays_out_dev || n_poses < 1 || H < 1 || W < 1 || row_begin < 0 || row_end > H || row_begin > row_end)
        return fail(c, NWE_ERR_INVALID, "bad pose / image / row range");
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    nwe_ctx::Slot* slot = nullptr;
    int rc = acquire_slot(c, &slot, false);
    if (rc) return rc;
    rc = upload_poses(c, *slot, c2w, n_poses, (hipStream_t)stream);
    if (rc) return rc;
    RenderArgs a = {};
    a.poses = slot->d_poses;
    a.H = H; a.W = W; a.row_begin = row_begin; a.rows = row_end - row_begin;
    a.n_rays = (int64_t)n_poses * a.rows * W;
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy; a.near = near; a.far = far;
    if (a.n_rays == 0) return NWE_OK;
    HIPCHK(c, hipEventRecord(slot->ev0, (hipStream_t)stream));
    hipLaunchKernelGGL(create_rays_kernel, dim3((unsigned)((a.n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, rays_out_dev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(slot->ev1, (hipStream_t)stream));
    slot->used = true;
    return NWE_OK;
}

int nwe_render_rays(nwe_ctx* c, const float* rays_dev, int64_t n_rays, int precision, const nwe_outputs* out, void* stream) {
    int rc = check_ready(c, out, precision);
    if (rc) return rc;
    if (n_rays < 0 || n_rays > INT32_MAX || (!rays_dev && n_rays > 0)) return fail(c, NWE_ERR_INVALID, "bad rays (null, or more than 2^31 - 1)");
    if (n_rays == 0) {
        c->dbg_z_fine = c->dbg_raw_c = c->dbg_raw_f = c->dbg_w = nullptr;
        c->trn_t = c->trn_nc = c->trn_nf = c->trn_u = nullptr;
        return NWE_OK;
    }
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    nwe_ctx::Slot* slot = nullptr;
    rc = acquire_slot(c, &slot);
    if (rc) return rc;
    RenderArgs a = {};
    a.rays = rays_dev; a.n_rays = n_rays; a.W = 1; a.rows = 1;
    a.ray_cols = c->net[0].in_dir == 0 ? 8 : 11;                       // rays.py:22-30: no view-direction columns without view dirs
    a.z_fine_in = c->dbg_z_fine; a.raw_in_c = c->dbg_raw_c; a.raw_in_f = c->dbg_raw_f; a.w_in = c->dbg_w;
    c->dbg_z_fine = c->dbg_raw_c = c->dbg_raw_f = c->dbg_w = nullptr;
    a.t_rand = c->trn_t; a.noise_c = c->trn_nc; a.noise_f = c->trn_nf; a.u_rand = c->trn_u;
    c->trn_t = c->trn_nc = c->trn_nf = c->trn_u = nullptr;
    a.out = *out;
    return launch(c, *slot, a, precision, stream);
}

int nwe_to8b(nwe_ctx* c, const float* rgb_dev, uint8_t* out_dev, int64_t n, void* stream) {
    if (!c || c->host_only || !rgb_dev || !out_dev || n < 0) return fail(c, NWE_ERR_INVALID, "bad argument");
    if (n == 0) return NWE_OK;
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(to8b_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb_dev, out_dev, n);
    HIPCHK(c, hipGetLastError());
    return NWE_OK;
}

int64_t nwe_flops_per_eval(const nwe_ctx* c, int which) {
    if (!c || which < 0 || which > 1 || !c->net[which].set) return 0;
    return c->net[which].flops;
}

float nwe_last_kernel_ms(nwe_ctx* c) {
    if (!c || c->host_only || c->last_slot < 0 || !c->slots[c->last_slot].used) return -1.f;
    const nwe_ctx::Slot& s = c->slots[c->last_slot];
    DeviceGuard guard;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipEventSynchronize(s.ev1);
    float ms = -1.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, s.ev0, s.ev1);
    if (e != hipSuccess) { c->err = std::string("nwe_last_kernel_ms: ") + hipGetErrorString(e); (void)hipGetLastError(); return -1.f; }
    return ms;
}

int nwe_last_launch_parts(nwe_ctx* c, float* ms2, int64_t* rays2) {
    if (!c || !ms2 || !rays2) return NWE_ERR_INVALID;
    ms2[0] = ms2[1] = -1.f; rays2[0] = rays2[1] = 0;
    if (c->host_only || c->last_slot < 0 || !c->slots[c->last_slot].used) return fail(c, NWE_ERR_STATE, "nothing has been launched");
    const nwe_ctx::Slot& s = c->slots[c->last_slot];
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(s.ev1));
    rays2[0] = s.rays_first; rays2[1] = s.rays_total - s.rays_first;
    if (s.has_mid) {
        HIPCHK(c, hipEventElapsedTime(&ms2[0], s.ev0, s.ev_mid));
        HIPCHK(c, hipEventElapsedTime(&ms2[1], s.ev_mid, s.ev1));
    } else {
        HIPCHK(c, hipEventElapsedTime(&ms2[0], s.ev0, s.ev1));
    }
    return NWE_OK;
}

int64_t nwe_packed_bytes(const nwe_ctx* c, int which) {
    if (!c || which < 0 || which > 1 || !c->net[which].set) return 0;
    return (int64_t)c->net[which].stream.size();
}

int nwe_packed_copy(const nwe_ctx* c, int which, void* host_dst, int64_t bytes) {
    if (!c || which < 0 || which > 1 || !host_dst || !c->net[which].set) return NWE_ERR_INVALID;
    if (bytes != (int64_t)c->net[which].stream.size()) return NWE_ERR_INVALID;
    std::memcpy(host_dst, c->net[which].stream.data(), (size_t)bytes);
    return NWE_OK;
}

int64_t nwe_packed_bias_count(const nwe_ctx* c, int which) {
    if (!c || which < 0 || which > 1 || !c->net[which].set) return 0;
    return (int64_t)c->net[which].bias_tab.size();
}

int nwe_packed_bias_copy(const nwe_ctx* c, int which, float* host_dst, int64_t count) {
    if (!c || which < 0 || which > 1 || !host_dst || !c->net[which].set) return NWE_ERR_INVALID;
    if (count != (int64_t)c->net[which].bias_tab.size()) return NWE_ERR_INVALID;
    std::memcpy(host_dst, c->net[which].bias_tab.data(), (size_t)count * sizeof(float));
    return NWE_OK;
}

float nwe_packed_scale(const nwe_ctx* c, int which) {
    if (!c || which < 0 || which > 1 || !c->net[which].set) return 0.f;
    return c->net[which].w_scale;
}

int nwe_debug_set_fine_depths(nwe_ctx* c, const float* z_dev) {
    if (!c) return NWE_ERR_INVALID;
    c->dbg_z_fine = z_dev;
    return NWE_OK;
}

int nwe_debug_set_raw(nwe_ctx* c, const float* raw_coarse_dev, const float* raw_fine_dev) {
    if (!c) return NWE_ERR_INVALID;
    c->dbg_raw_c = raw_coarse_dev; c->dbg_raw_f = raw_fine_dev;
    return NWE_OK;
}

int nwe_debug_set_coarse_weights(nwe_ctx* c, const float* weights_dev) {
    if (!c) return NWE_ERR_INVALID;
    c->dbg_w = weights_dev;
    return NWE_OK;
}

int nwe_debug_set_fold(nwe_ctx* c, int on) {
    if (!c) return NWE_ERR_INVALID;
    c->fold = on ? 1 : 0;
    return NWE_OK;
}

int nwe_debug_set_decomposition(nwe_ctx* c, int mode) {
    if (!c || mode < -1 || mode > 2) return NWE_ERR_INVALID;
    c->decomposition = mode;
    return NWE_OK;
}

int nwe_debug_last_plan(const nwe_ctx* c) { return c ? c->last_plan : -1; }

int nwe_debug_peer_access(const nwe_ctx* first, const nwe_ctx* tile) {
    if (!first || !tile || first->host_only || tile->host_only) return -1;
    if (tile->peer_access != -2) return tile->peer_access;
    if (tile->device == first->device) return 1;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, tile->device, first->device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return can ? 1 : 0;
}

const char* nwe_last_warning(const nwe_ctx* c) { return c ? c->warn.c_str() : ""; }

int nwe_debug_set_stamps(nwe_ctx* c, unsigned long long* per_wave_dev) {
    if (!c) return NWE_ERR_INVALID;
    c->stamps = per_wave_dev;
    return NWE_OK;
}

int nwe_set_white_background(nwe_ctx* c, int on) {
    if (!c) return NWE_ERR_INVALID;
    c->white_bkgd = on ? 1 : 0;
    return NWE_OK;
}

int nwe_set_train_tables(nwe_ctx* c, const float* t_rand_dev, const float* noise_coarse_dev, const float* noise_fine_dev,
                         const float* u_sorted_dev) {
    if (!c) return NWE_ERR_INVALID;
    c->trn_t = t_rand_dev; c->trn_nc = noise_coarse_dev; c->trn_nf = noise_fine_dev; c->trn_u = u_sorted_dev;
    return NWE_OK;
}

int nwe_selftest(nwe_ctx* c, int32_t* report8) {
    if (!c || c->host_only || !report8) return fail(c, NWE_ERR_INVALID, "bad argument");
    DeviceGuard guard;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = run_selftest(report8, nullptr);
    if (rc == -1) return fail(c, NWE_ERR_HIP, "selftest: HIP failure");
    if (rc != 0) return fail(c, NWE_ERR_STATE, "selftest: a hardware layout assumption does not hold (see report)");
    return NWE_OK;
}

}  // extern "C"
