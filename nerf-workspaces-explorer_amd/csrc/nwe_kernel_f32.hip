// fp32 render kernel (NWE_PREC_F32): the whole ray pipeline of
// nerf/inference/nerf_replica_inference_handler.py:203-277 with the MLP GEMMs as fp32 FMA chains on
// the vector ALU.  Any layer shape with width <= 256.  It is the on-device reference the MFMA kernel
// is compared with at full frame size (where the CPU oracle needs minutes), and the path for network
// shapes the MFMA kernel has no instantiation for.
//
// One 256-thread workgroup owns a packet of 16 rays and walks their samples in lock step: thread n is
// output neuron n of the current layer and keeps 16 accumulators (one per ray of the packet), the
// layer input lives in LDS as [k][16] so one ds_read_b128 broadcast feeds 4 FMAs.  Weights are read
// transposed ([k][n], coalesced, L2-resident).
#include "nwe_device.h"
#include "nwe_host.h"

namespace nwe {

constexpr int kRP = 16;  // rays per workgroup

__device__ __forceinline__ void dense16(const float* __restrict__ blob, const LayerF32& L, const float* in1, int K1,
                                        const float* in2, int K2, float* out, bool relu) {
    const int n = threadIdx.x;
    if (n < L.N) {
        const float* wt = blob + L.wt_off;
        float acc[kRP];
        const float bias = blob[L.b_off + n];
#pragma unroll
        for (int p = 0; p < kRP; ++p) acc[p] = bias;
        for (int k = 0; k < K1; ++k) {
            const float w = wt[(int64_t)k * L.N + n];
            const float4* h = reinterpret_cast<const float4*>(in1 + k * kRP);
#pragma unroll
            for (int q = 0; q < kRP / 4; ++q) {
                const float4 v = h[q];
                acc[4 * q + 0] = __fmaf_rn(w, v.x, acc[4 * q + 0]);
                acc[4 * q + 1] = __fmaf_rn(w, v.y, acc[4 * q + 1]);
                acc[4 * q + 2] = __fmaf_rn(w, v.z, acc[4 * q + 2]);
                acc[4 * q + 3] = __fmaf_rn(w, v.w, acc[4 * q + 3]);
            }
        }
        for (int k = 0; k < K2; ++k) {
            const float w = wt[(int64_t)(K1 + k) * L.N + n];
            const float4* h = reinterpret_cast<const float4*>(in2 + k * kRP);
#pragma unroll
            for (int q = 0; q < kRP / 4; ++q) {
                const float4 v = h[q];
                acc[4 * q + 0] = __fmaf_rn(w, v.x, acc[4 * q + 0]);
                acc[4 * q + 1] = __fmaf_rn(w, v.y, acc[4 * q + 1]);
                acc[4 * q + 2] = __fmaf_rn(w, v.z, acc[4 * q + 2]);
                acc[4 * q + 3] = __fmaf_rn(w, v.w, acc[4 * q + 3]);
            }
        }
        float4* o = reinterpret_cast<float4*>(out + n * kRP);
#pragma unroll
        for (int q = 0; q < kRP / 4; ++q) {
            float4 v = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            o[q] = v;
        }
    }
    __syncthreads();
}

// gamma(v) rows [f][16]: nerf/models/embedding.py:24-48.  f = 0..2 identity, then per band sin(3), cos(3).
__device__ __forceinline__ void encode16(float* dst, const float* src3 /*[3][16]*/, int n_feat, float inv_div) {
    const int p = threadIdx.x & (kRP - 1);
    for (int f = threadIdx.x / kRP; f < n_feat; f += 256 / kRP) {
        float val;
        if (f < 3) {
            val = __fdiv_rn(src3[f * kRP + p], inv_div);
        } else {
            const int q = f - 3, band = q / 6, within = q % 6, c = within % 3;
            const float v = __fdiv_rn(src3[c * kRP + p], inv_div);                   // embedding.py:48
            const float arg = __fmul_rn(v, (float)(1 << band));                       // exact (power of two)
            val = within >= 3 ? cosf(arg) : sinf(arg);
        }
        dst[f * kRP + p] = val;
    }
}

__global__ void __launch_bounds__(256) render_f32_kernel(RenderArgs a, NetF32 nc, NetF32 nf) {
    __shared__ __attribute__((aligned(16))) float s_gx[96 * kRP];
    __shared__ __attribute__((aligned(16))) float s_gd[64 * kRP];
    __shared__ __attribute__((aligned(16))) float s_ha[256 * kRP];
    __shared__ __attribute__((aligned(16))) float s_hb[256 * kRP];
    __shared__ __attribute__((aligned(16))) float s_pt[3 * kRP];
    __shared__ __attribute__((aligned(16))) float s_raw[4 * kRP];
    __shared__ float s_wcur[kRP];   // endpoint_feat: the weight of the current sample, per ray of the packet
    __shared__ float s_w[kMaxSamples * kRP];
    __shared__ float s_t[kMaxSamples], s_omt[kMaxSamples], s_u[kMaxImportance];

    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kRP;
    const bool owner = tid < kRP;
    const int64_t ridx = base + tid;
    const bool live = owner && ridx < a.n_rays;
    const int ns = a.n_samples, ni = a.n_importance;

    for (int i = tid; i < ns; i += 256) { s_t[i] = a.t_vals[i]; s_omt[i] = a.omt_vals[i]; }
    for (int i = tid; i < ni; i += 256) s_u[i] = a.u_vals[i];

    Ray ray = {};
    if (owner) {
        ray = load_ray(a, live ? ridx : a.n_rays - 1);
        s_pt[0 * kRP + tid] = ray.vx; s_pt[1 * kRP + tid] = ray.vy; s_pt[2 * kRP + tid] = ray.vz;
    }
    __syncthreads();
    encode16(s_gd, s_pt, nc.in_dir, 1.f);   // handler.py:101 scalar_factor = 1; same for every sample (model_utils.py:23-25)
    __syncthreads();

    Composite comp; comp.reset();
    FineSampler fs;
    fs.wc = s_w + tid; fs.stride = kRP; fs.u_tab = s_u; fs.ns = ns; fs.ni = ni;
    fs.cd.t_tab = s_t; fs.cd.omt_tab = s_omt; fs.cd.ns = ns;
    const int64_t rrow = live ? ridx : a.n_rays - 1;
    fs.cd.jitter = a.t_rand; fs.cd.row = (int)rrow;
    fs.u_rand = a.u_rand;
    uint32_t flags = 0;

    for (int pass = 0; pass < (ni > 0 ? 2 : 1); ++pass) {
        const NetF32& net = pass == 0 ? nc : nf;
        const int S = pass == 0 ? ns : ns + ni;
        const float* raw_in = pass == 0 ? a.raw_in_c : a.raw_in_f;          // test hook: network outputs from the caller
        if (pass == 0 && a.w_in) {                                          // test hook: coarse weights from the caller
            if (owner) for (int s = 0; s < ns; ++s) s_w[s * kRP + tid] = a.w_in[rrow * ns + s];
            continue;
        }
        const bool want_feat = pass == 1 && a.out.feat_map != nullptr && !raw_in;   // uniform
        float facc[kRP];
#pragma unroll
        for (int p = 0; p < kRP; ++p) facc[p] = 0.f;
        const float* s_feat = s_ha;
        float z_cur = 0.f, z_next = 0.f;
        if (owner) {
            comp.reset();
            if (pass == 0) { z_cur = fs.cd.z(ray, 0); }
            else {
                fs.prepare(ray);
                if (wants_survey(a.out)) {
                    const SampleSurvey sv = fs.survey(ray);
                    if (live) flags |= store_survey(a.out, ridx, sv);
                }
                z_cur = a.z_fine_in ? a.z_fine_in[(live ? ridx : a.n_rays - 1) * S] : fs.next(ray);
            }
        }
        for (int s = 0; s < S; ++s) {
            if (owner) {
                if (s + 1 < S) {
                    if (pass == 0) z_next = fs.cd.z(ray, s + 1);
                    else z_next = a.z_fine_in ? a.z_fine_in[(live ? ridx : a.n_rays - 1) * S + s + 1] : fs.next(ray);
                }
                float px, py, pz; point_at(ray, z_cur, px, py, pz);
                s_pt[0 * kRP + tid] = px; s_pt[1 * kRP + tid] = py; s_pt[2 * kRP + tid] = pz;
            }
            __syncthreads();
            if (raw_in) {
                if (owner) {
                    const float4 v = *reinterpret_cast<const float4*>(raw_in + (rrow * S + s) * 4);
                    s_raw[0 * kRP + tid] = v.x; s_raw[1 * kRP + tid] = v.y; s_raw[2 * kRP + tid] = v.z; s_raw[3 * kRP + tid] = v.w;
                }
            } else {
            encode16(s_gx, s_pt, net.in_xyz, 10.f);   // handler.py:93 scalar_factor = 10
            __syncthreads();
            // trunk: nerf_model.py:53-59
            float* cur = s_ha; float* nxt = s_hb;
            dense16(net.blob, net.pts[0], s_gx, net.in_xyz, nullptr, 0, cur, true);
            for (int i = 1; i < net.D; ++i) {
                if (i == net.skip + 1) dense16(net.blob, net.pts[i], s_gx, net.in_xyz, cur, net.W, nxt, true);
                else dense16(net.blob, net.pts[i], cur, net.W, nullptr, 0, nxt, true);
                float* t = cur; cur = nxt; nxt = t;
            }
            if (net.in_dir == 0) {
                // use_view_dirs=False, nerf_model.py:82-83: outputs = _output_linear(h); channels 0..2 rgb_raw, 3 sigma_raw
                // (model_utils.py:62,71), the rest unused.  out_ch rows land in nxt, the four that matter are copied.
                dense16(net.blob, net.output, cur, net.W, nullptr, 0, nxt, false);
                if (tid < 4 * kRP) s_raw[tid] = nxt[tid];
                __syncthreads();
            } else {
            // heads: nerf_model.py:63-74.  alpha (1 row) goes to s_raw row 3, feature to nxt, views to cur, rgb to s_raw rows 0..2
            dense16(net.blob, net.alpha, cur, net.W, nullptr, 0, s_raw + 3 * kRP, false);
            dense16(net.blob, net.feature, cur, net.W, nullptr, 0, nxt, false);
            dense16(net.blob, net.views, nxt, net.W, s_gd, net.in_dir, cur, true);
            dense16(net.blob, net.rgb, cur, net.W / 2, nullptr, 0, s_raw, false);
            s_feat = cur;                         // the view layer's output [W/2][16]: the endpoint feature (nerf_model.py:72-73)
            }
            }
            if (owner) {
                const float rr = s_raw[0 * kRP + tid], rg = s_raw[1 * kRP + tid], rb = s_raw[2 * kRP + tid],
                            rs = s_raw[3 * kRP + tid];
                const float* nz = pass == 0 ? a.noise_c : a.noise_f;
                const float w = comp.step(rr, rg, rb, rs, z_cur, z_next, s + 1 == S, ray.dnorm, nz ? nz[rrow * S + s] : 0.f);
                if (pass == 0) s_w[s * kRP + tid] = w;
                if (want_feat) s_wcur[tid] = w;
                if (live) {
                    if (pass == 0 && a.out.weights_coarse) a.out.weights_coarse[ridx * S + s] = w;
                    float* raw = pass == 0 ? a.out.raw_coarse : a.out.raw_fine;
                    if (raw) {
                        float4* dst = reinterpret_cast<float4*>(raw + (ridx * S + s) * 4);
                        *dst = make_float4(rr, rg, rb, rs);
                        if (bad(rr) || bad(rg) || bad(rb) || bad(rs)) flags |= NWE_FLAG_RAW;
                    }
                    if (pass == 1 && a.out.z_fine) a.out.z_fine[ridx * S + s] = z_cur;
                }
                z_cur = z_next;
            }
            if (want_feat) {
                // feat_map = sum_s weights * feat (model_utils.py:87-89): thread n owns channel n for the 16 rays of the packet.
                // s_feat is rewritten by the next sample's trunk only behind a barrier every thread passes after this.
                __syncthreads();
                if (tid < net.W / 2) {
#pragma unroll
                    for (int p = 0; p < kRP; ++p) facc[p] = __fadd_rn(facc[p], __fmul_rn(s_wcur[p], s_feat[tid * kRP + p]));
                }
            }
            // s_pt / s_raw are rewritten only after the next barrier pair; the owners' reads above are
            // ordered before their own writes at the top of the next iteration.
        }
        if (want_feat && tid < net.W / 2) {
            for (int p = 0; p < kRP; ++p)
                if (base + p < a.n_rays) a.out.feat_map[(base + p) * (net.W / 2) + tid] = facc[p];
        }
        if (live) {
            flags |= store_ray(a.out, ridx, comp, pass == 1, a.white_bkgd != 0);
            if (ni == 0) flags |= store_ray(a.out, ridx, comp, true, a.white_bkgd != 0);   // coarse-only: fill the "fine" slots too
        }
    }
    if (flags && a.out.flags) atomicOr(a.out.flags, flags);
}

void launch_render_f32(const RenderArgs& a, const NetF32& nc, const NetF32& nf, hipStream_t stream) {
    const int64_t blocks = (a.n_rays + kRP - 1) / kRP;
    hipLaunchKernelGGL(render_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, nc, nf);
}

}  // namespace nwe
