// Device-side building blocks shared by the fp32 and the MFMA render kernels: ray set-up, coarse
// depths, front-to-back compositing, inverse-CDF fine sampling with the sorted merge.
//
// Every formula follows the reference's op ORDER (file:line relative to the reference root), because
// the top positional-encoding band multiplies a point coordinate by 2^9/10: one fp32 ulp of the point
// is ~5e-5 rad there (SURVEY.md §7 hard part 3).  Hence: no FMA contraction where torch rounds twice
// (this translation unit is compiled with -ffp-contract=off and uses __f*_rn where it matters), true
// divisions, and the fp64 running product/sum that torch's CPU cumprod/cumsum use (measured: their
// accumulator is double, rounded to float at every element).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nwe.h"

namespace nwe {

constexpr int kMaxSamples = 128;     // n_samples upper bound (LDS tables)
constexpr int kMaxImportance = 256;  // n_importance upper bound

// Kernel arguments common to both kernels (passed by value).
struct RenderArgs {
    // ray source: either precomputed rays [n_rays,11] or pinhole poses
    const float* rays;       // device, may be null -> generate from poses
    int ray_cols;            // columns of `rays`: 11, or 8 for networks without view directions (rays.py:22-30)
    const float* poses;      // device, n_poses x 16 (row-major c2w)
    int64_t n_rays;          // total rays of the call
    int64_t ray_first;       // first ray of THIS launch (a call may be split into launches with different decompositions)
    int H, W, row_begin, rows;  // rows = row_end-row_begin; rays per pose = rows*W
    float fx, fy, cx, cy, near, far;
    // sampling tables (device): t[ns], 1-t[ns], u[ni]
    const float* t_vals;
    const float* omt_vals;
    const float* u_vals;
    const float* z_fine_in;  // test hook: fine depths [n_rays, ns+ni] instead of importance sampling
    const float* raw_in_c;   // test hook: network outputs [n_rays, ns, 4] of the coarse pass instead of evaluating the MLP
    const float* raw_in_f;   // test hook: ... [n_rays, ns+ni, 4] of the fine pass
    const float* w_in;       // test hook: coarse weights [n_rays, ns] instead of the coarse pass (its outputs are not written)
    // training-mode forward (nerf/training/nerf_replica_training_handler.py:553-580): random numbers drawn by the host
    // where the reference calls torch.rand / torch.randn, one row per ray of this call; each may be null (= inference)
    const float* t_rand;     // [n_rays, ns]      stratified jitter in [0,1)                 (:560-562)
    const float* noise_c;    // [n_rays, ns]      sigma noise, already times raw_noise_std   (model_utils.py:64-71)
    const float* noise_f;    // [n_rays, ns+ni]
    const float* u_rand;     // [n_rays, ni]      inverse-CDF arguments, ASCENDING per ray   (rays.py:98)
    int white_bkgd;          // rendering.white_background (model_utils.py:97-98): rgb += 1 - acc
    int n_samples, n_importance;
    unsigned long long* stamps;  // diagnostic builds (-DNWE_STAMPS): per-wave cycle sums, else unused
    nwe_outputs out;
};

struct Ray {
    float ox, oy, oz, dx, dy, dz, near, far, vx, vy, vz, dnorm;
};

// sqrt(x^2+y^2+z^2) exactly as torch.norm(dim=-1) evaluates a 3-vector on CPU: an FMA chain over the
// elements in order (measured bit-exact on 20k vectors), correctly rounded sqrt.
__device__ __forceinline__ float norm3(float x, float y, float z) {
    return __builtin_sqrtf(__fmaf_rn(z, z, __fmaf_rn(y, y, __fmul_rn(x, x))));   // __fsqrt_rn is the NATIVE (1-ulp) sqrt in HIP
}

// A ray is kept between uses as its SEED - three registers - and expanded where it is needed (make_ray): the render
// kernels' per-ray state must not stay in registers across an MLP evaluation, which needs all of them.
struct RaySeed {
    int pose;     // pose index (pinhole mode) / row of the ray table (precomputed rays)
    float x, y;   // camera-space direction ((w-cx)/fx, (h-cy)/fy, 1), pinhole mode only
};

// nerf/rays/rays.py:35-58.  idx is the ray index inside the call.
__device__ __forceinline__ RaySeed seed_ray(const RenderArgs& a, int64_t idx) {
    RaySeed s;
    if (a.rays) { s.pose = (int)idx; s.x = s.y = 0.f; return s; }
    const int per_pose = a.rows * a.W;
    s.pose = (int)(idx / per_pose);
    const int rem = (int)(idx - (int64_t)s.pose * per_pose);
    const int h = a.row_begin + rem / a.W, w = rem % a.W;
    // rays.py:52-56: ((w-cx)/fx, (h-cy)/fy, 1), true divisions
    s.x = __fdiv_rn(__fsub_rn((float)w, a.cx), a.fx);
    s.y = __fdiv_rn(__fsub_rn((float)h, a.cy), a.fy);
    return s;
}

// nerf/rays/rays.py:6-32, :61-71.  VIEW: also the normalised view direction (rays.py:24), needed once per ray for gamma(d).
template <bool VIEW>
__device__ __forceinline__ Ray make_ray(const RenderArgs& a, const RaySeed& s) {
    Ray r;
    r.vx = r.vy = r.vz = 0.f;
    if (a.rays) {  // handler.py:210-214: columns [o d near far viewdir]
        const float* p = a.rays + (int64_t)s.pose * a.ray_cols;
        r.ox = p[0]; r.oy = p[1]; r.oz = p[2];
        r.dx = p[3]; r.dy = p[4]; r.dz = p[5];
        r.near = p[6]; r.far = p[7];
        if (VIEW && a.ray_cols > 8) { r.vx = p[8]; r.vy = p[9]; r.vz = p[10]; }
    } else {
        const float* m = a.poses + s.pose * 16;
        // rays.py:67: 3x3 @ 3x1 as torch's CPU bmm does it: products summed left to right, no FMA
        r.dx = __fadd_rn(__fadd_rn(__fmul_rn(m[0], s.x), __fmul_rn(m[1], s.y)), m[2]);
        r.dy = __fadd_rn(__fadd_rn(__fmul_rn(m[4], s.x), __fmul_rn(m[5], s.y)), m[6]);
        r.dz = __fadd_rn(__fadd_rn(__fmul_rn(m[8], s.x), __fmul_rn(m[9], s.y)), m[10]);
        r.ox = m[3]; r.oy = m[7]; r.oz = m[11];          // rays.py:68
        r.near = a.near; r.far = a.far;                  // rays.py:26
        if (VIEW) {
            const float n = norm3(r.dx, r.dy, r.dz);     // rays.py:24
            r.vx = __fdiv_rn(r.dx, n); r.vy = __fdiv_rn(r.dy, n); r.vz = __fdiv_rn(r.dz, n);
        }
    }
    r.dnorm = norm3(r.dx, r.dy, r.dz);                   // model_utils.py:60
    return r;
}

__device__ __forceinline__ Ray load_ray(const RenderArgs& a, int64_t idx) { return make_ray<true>(a, seed_ray(a, idx)); }

// handler.py:218: near*(1-t) + far*t, two products and one sum
__device__ __forceinline__ float coarse_z(const Ray& r, float t, float omt) {
    return __fadd_rn(__fmul_rn(r.near, omt), __fmul_rn(r.far, t));
}

// Coarse depth i of a ray: the linspace depth (inference), or, with a row of stratified-jitter numbers, a point of the
// interval between the mid points around it (training_handler.py:553-562): mids = .5*(z[1:] + z[:-1]),
// upper = [mids, z[-1]], lower = [z[0], mids], z = lower + (upper - lower) * t_rand.
struct CoarseDepths {
    const float* t_tab; const float* omt_tab;   // LDS tables t, 1-t
    const float* jitter;                        // t_rand [n_rays, ns] (global, wave-uniform pointer) or null
    int row;                                    // this ray's row in the per-ray tables of the call
    int ns;
    __device__ __forceinline__ float base(const Ray& r, int i) const { return coarse_z(r, t_tab[i], omt_tab[i]); }
    __device__ __forceinline__ float z(const Ray& r, int i) const {
        const float zi = base(r, i);
        if (!jitter) return zi;
        const float lower = i > 0 ? __fmul_rn(.5f, __fadd_rn(zi, base(r, i - 1))) : zi;
        const float upper = i + 1 < ns ? __fmul_rn(.5f, __fadd_rn(base(r, i + 1), zi)) : zi;
        return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), jitter[(int64_t)row * ns + i]));
    }
};

// handler.py:223 / :246: o + d*z, product then sum (no FMA)
__device__ __forceinline__ void point_at(const Ray& r, float z, float& px, float& py, float& pz) {
    px = __fadd_rn(r.ox, __fmul_rn(r.dx, z));
    py = __fadd_rn(r.oy, __fmul_rn(r.dy, z));
    pz = __fadd_rn(r.oz, __fmul_rn(r.dz, z));
}

// Front-to-back alpha compositing of one ray, one sample per step.  nerf/models/model_utils.py:49-100.
struct Composite {
    double t_run;  // running product of (1-alpha+1e-10), fp64 like torch's CPU cumprod (:79)
    float r, g, b, depth, acc;
    __device__ __forceinline__ void reset() {
        t_run = 1.0; r = g = b = depth = acc = 0.f;
    }
    // The per-sample half that needs no running state (model_utils.py:49-62): opacity and colour of one sample from the
    // network output, its depth and the next depth (ignored when last).
    static __device__ __forceinline__ float4 shade(float raw_r, float raw_g, float raw_b, float raw_s, float z, float z_next,
                                                   bool last, float dnorm, float noise = 0.f) {
        float dist = last ? 1e10f : __fsub_rn(z_next, z);                    // :51-56
        dist = __fmul_rn(dist, dnorm);                                       // :60
        const float sig = fmaxf(__fadd_rn(raw_s, noise), 0.f);               // relu(raw + noise), :49,:71 (noise = 0. in inference)
        const float alpha = __fsub_rn(1.f, expf(-__fmul_rn(sig, dist)));     // :49
        const float cr = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-raw_r)));       // sigmoid, :62
        const float cg = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-raw_g)));
        const float cb = __fdiv_rn(1.f, __fadd_rn(1.f, expf(-raw_b)));
        return make_float4(cr, cg, cb, alpha);
    }
    // The sequential half (:79-95), in sample order: returns the sample's weight (alpha * transmittance).
    __device__ __forceinline__ float accumulate(float4 ca, float z) {
        const float alpha = ca.w;
        const float w = __fmul_rn(alpha, (float)t_run);                      // :79-80 (cumprod shifted by one)
        t_run *= (double)__fadd_rn(__fsub_rn(1.f, alpha), 1e-10f);
        r = __fadd_rn(r, __fmul_rn(w, ca.x));                                // :84
        g = __fadd_rn(g, __fmul_rn(w, ca.y));
        b = __fadd_rn(b, __fmul_rn(w, ca.z));
        depth = __fadd_rn(depth, __fmul_rn(w, z));                           // :93
        acc = __fadd_rn(acc, w);                                             // :95
        return w;
    }
    // raw = network output for this sample, z = its depth, z_next = next depth (ignored when last).
    __device__ __forceinline__ float step(float raw_r, float raw_g, float raw_b, float raw_s, float z, float z_next,
                                          bool last, float dnorm, float noise = 0.f) {
        return accumulate(shade(raw_r, raw_g, raw_b, raw_s, z, z_next, last, dnorm, noise), z);
    }
    // :94  1 / max(1e-10, depth/acc); torch.max propagates NaN (acc == 0 -> NaN)
    __device__ __forceinline__ float disp() const {
        const float q = __fdiv_rn(depth, acc);
        const float m = (q != q) ? q : fmaxf(1e-10f, q);
        return __fdiv_rn(1.f, m);
    }
};

// Inverse-CDF importance sampling merged with the coarse depths, one depth per call in sorted order.
// nerf/rays/rays.py:74-121 (det=True) + handler.py:236-243.  The coarse depths and the importance
// samples are each already sorted, so torch.sort(cat(...)) is a two-way merge of the same values.
// `wc` points at this ray's coarse weights in LDS, element i at wc[i*stride]; prepare() overwrites
// elements 0..ns-2 with the cdf (rays.py:87-90).
// What the importance samples of a ray look like as a whole: z_std (handler.py:267) and the conditioning diagnostics of
// include/nwe.h.  A pure function of the cdf, computed once per ray by FineSampler::survey() and stored at once, so that
// none of it is carried through the fine pass.
struct SampleSurvey {
    float z_std;
    float min_denom;   // smallest cdf step BEFORE the < 1e-5 -> 1 replacement (samples between two different cdf entries)
    float max_amp;     // largest bin width / cdf step used
    float min_switch;  // smallest |cdf step - 1e-5|
};

struct FineSampler {
    float* wc; int stride;
    CoarseDepths cd;
    const float* u_tab;      // LDS table of the deterministic u (rays.py:95)
    const float* u_rand;     // ascending random u [n_rays, ni] (rays.py:98, global, wave-uniform pointer) or null; row = cd.row
    int ns, ni;
    int ci, fj, ptr;
    float cur_f;

    __device__ __forceinline__ float zc(const Ray& r, int i) const { return cd.z(r, i); }
    __device__ __forceinline__ float zmid(const Ray& r, int k) const {       // handler.py:236
        return __fmul_rn(.5f, __fadd_rn(zc(r, k + 1), zc(r, k)));
    }
    // torch.sum(weights, -1) (rays.py:88) in the order torch's CPU kernel adds a contiguous fp32 row on x86
    // (aten/src/ATen/native/cpu/SumKernel.cpp: vectorized_inner_sum / row_sum; sum_stub has no AVX-512 registration, so the
    // 8-lane kernel runs on every AVX2-or-later machine): the row is cut into 8-wide vectors; vector 4g+k goes to partial
    // accumulator k (k < 4, while whole groups of four last), the vectors left over to accumulator 0, then accumulators 1..3
    // are added to 0; the scalar result starts with the elements behind the last whole vector and then adds the eight
    // lanes.  The plain left-to-right sum differs from it by up to 21 ulp on the benchmark scene, which the inverse CDF of
    // a nearly empty bin amplifies into a visible depth shift; with this order the cdf is torch's bit for bit.
    // Rows shorter than one vector take the scalar kernel (scalar_inner_sum): the same four interleaved accumulators
    // over single elements.  Element i is weights[i + 1] + 1e-5 (rays.py:87).  n <= 126 keeps the cascade levels of
    // multi_row_sum out of play.  (tests/test_host_logic.py restates this in numpy and checks it against torch.sum.)
    __device__ __forceinline__ float torch_sum_order(int n) const {
        if (n < 8) {
            float p[4] = {0.f, 0.f, 0.f, 0.f};
            const int ng4 = n >> 2;
            for (int g = 0; g < ng4; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = __fadd_rn(p[k], __fadd_rn(wc[(4 * g + k + 1) * stride], 1e-5f));
            for (int i = 4 * ng4; i < n; ++i) p[0] = __fadd_rn(p[0], __fadd_rn(wc[(i + 1) * stride], 1e-5f));
            return __fadd_rn(__fadd_rn(__fadd_rn(p[0], p[1]), p[2]), p[3]);
        }
        const int nv = n >> 3, ng = nv >> 2;
        float acc = 0.f;
        for (int k = 8 * nv; k < n; ++k) acc = __fadd_rn(acc, __fadd_rn(wc[(k + 1) * stride], 1e-5f));
        for (int l = 0; l < 8; ++l) {
            float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
            for (int g = 0; g < ng; ++g) {
                const int e = 32 * g + l + 1;
                p0 = __fadd_rn(p0, __fadd_rn(wc[e * stride], 1e-5f));
                p1 = __fadd_rn(p1, __fadd_rn(wc[(e + 8) * stride], 1e-5f));
                p2 = __fadd_rn(p2, __fadd_rn(wc[(e + 16) * stride], 1e-5f));
                p3 = __fadd_rn(p3, __fadd_rn(wc[(e + 24) * stride], 1e-5f));
            }
            for (int i = 4 * ng; i < nv; ++i) p0 = __fadd_rn(p0, __fadd_rn(wc[(8 * i + l + 1) * stride], 1e-5f));
            p0 = __fadd_rn(__fadd_rn(__fadd_rn(p0, p1), p2), p3);
            acc = __fadd_rn(acc, p0);
        }
        return acc;
    }
    __device__ __forceinline__ void prepare(const Ray& r) { build_cdf(); start(r); }
    // weights[..., 1:-1] + 1e-5, normalised, cumulative (double accumulator, float per element), in place
    __device__ __forceinline__ void build_cdf() {
        const float sum = torch_sum_order(ns - 2);
        double run = 0.0;
        wc[0] = 0.f;                                                         // rays.py:90 leading zero
        for (int i = 1; i < ns - 1; ++i) {                                   // cdf[i] replaces weight i in place
            run += (double)__fdiv_rn(__fadd_rn(wc[i * stride], 1e-5f), sum);
            wc[i * stride] = (float)run;
        }
    }
    __device__ __forceinline__ void start(const Ray& r) {
        ci = 0; fj = 0; ptr = 0;
        cur_f = sample(r, 0, ptr, nullptr);
    }
    // Importance sample j (u ascending, so the searchsorted position `p` only moves forward); sv != null also records the
    // sample's conditioning.
    __device__ __forceinline__ float sample(const Ray& r, int j, int& p, SampleSurvey* sv) const {
        const int ncdf = ns - 1;
        const float u = u_rand ? u_rand[(int64_t)cd.row * ni + j] : u_tab[j];
        while (p < ncdf && wc[p * stride] <= u) ++p;                         // searchsorted(right=True), :103
        const int below = max(p - 1, 0), above = min(p, ncdf - 1);           // :104-105
        const float cb = wc[below * stride], ca = wc[above * stride];
        const float bb = zmid(r, below), ba = zmid(r, above);
        float denom = __fsub_rn(ca, cb);                                     // :113
        const float raw_denom = denom;
        if (denom < 1e-5f) denom = 1.f;                                      // :114
        const float width = __fsub_rn(ba, bb);
        if (sv) {
            if (above != below) {
                sv->min_denom = fminf(sv->min_denom, raw_denom);
                sv->min_switch = fminf(sv->min_switch, fabsf(raw_denom - 1e-5f));
                sv->max_amp = fmaxf(sv->max_amp, width / denom);
            } else if (ncdf >= 2) {
                // u at or above the last cdf entry (u = 1.0 against a cdf that ends at 1 - 1 ulp, :103-105): the sample sits on
                // the last bin edge.  Whether it does is itself a rounding matter - with the last entry one ulp higher the
                // sample is interpolated in the last bin - so that bin's step counts for the amplification.
                float dl = __fsub_rn(wc[(ncdf - 1) * stride], wc[(ncdf - 2) * stride]);
                if (dl < 1e-5f) dl = 1.f;
                sv->max_amp = fmaxf(sv->max_amp, __fsub_rn(zmid(r, ncdf - 1), zmid(r, ncdf - 2)) / dl);
            }
        }
        const float t = __fdiv_rn(__fsub_rn(u, cb), denom);                  // :118
        return __fadd_rn(bb, __fmul_rn(t, width));                           // :119
    }
    // All importance samples of the ray once over (after prepare()): std(z_samples, unbiased=False) (handler.py:267) and
    // the conditioning diagnostics.
    __device__ __forceinline__ SampleSurvey survey(const Ray& r) const {
        SampleSurvey sv = {0.f, 1.f, 0.f, 1.f};
        double s1 = 0.0, s2 = 0.0;
        int p = 0;
        for (int j = 0; j < ni; ++j) {
            const float z = sample(r, j, p, &sv);
            s1 += (double)z; s2 += (double)z * (double)z;
        }
        const double m = s1 / ni, v = s2 / ni - m * m;
        sv.z_std = (float)sqrt(v > 0.0 ? v : 0.0);
        return sv;
    }
    __device__ __forceinline__ float next(const Ray& r) {
        const float a = ci < ns ? zc(r, ci) : INFINITY;
        const float f = fj < ni ? cur_f : INFINITY;
        if (!(fj < ni) || a <= f) { ++ci; return a; }
        ++fj;
        if (fj < ni) cur_f = sample(r, fj, ptr, nullptr);
        return f;
    }
};

// sin and cos of a * 2^b for b = 0..NB-1, a = v * first (first a power of two, so a is exact): the NB octaves of one
// positional-encoding coordinate (nerf/models/embedding.py:36 evaluates sin/cos of the fp32 product x * 2^b, which is exact).
// One fp64 evaluation at the lowest octave - two-constant Cody-Waite reduction by pi/2 (exact for |a| < 1.6e6), the
// classic degree-13/12 minimax kernels on [-pi/4, pi/4] (error < 2^-57) - then double-angle steps in fp64, whose error
// doubles per octave and stays below 1e-13 after nine of them; every result is rounded to fp32 once.  This replaces
// 2*NB libm calls per coordinate (each with its own range reduction) and is at least as close to the reference's libm
// as they were (nwe_selftest report[4]: max |err| vs fp64 in units of 1e-9).
template <int NB>
__device__ __forceinline__ void octave_sincos(float v, float first, float* sn, float* cs) {
    const double a = (double)v * (double)first;
    const double n = __builtin_rint(a * 6.36619772367581382433e-01);         // 2/pi
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, a);              // first 33 bits of pi/2: the product is exact
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sin_r = __builtin_fma(r * z, ps, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cos_r = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)n & 3;                     // a = r + n*pi/2
    double s = (q & 1) ? cos_r : sin_r;
    double c = (q & 1) ? sin_r : cos_r;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sn[b] = (float)s;
        cs[b] = (float)c;
        if (b + 1 < NB) {
            const double t = s * c, u = s * s;
            s = t + t;                            // sin 2x = 2 sin x cos x
            c = __builtin_fma(-2.0, u, 1.0);      // cos 2x = 1 - 2 sin^2 x
        }
    }
}

__device__ __forceinline__ bool bad(float x) { return !(fabsf(x) <= 3.402823466e38f); }  // NaN or inf

// Writes the per-ray results of one pass into the fine (handler.py:263-266) or coarse (:257-260) slots
// and returns the NaN/Inf flag bits of what it wrote (handler.py:273-275).
__device__ __forceinline__ uint32_t store_ray(const nwe_outputs& o, int64_t idx, Composite c, bool fine, bool white_bkgd = false) {
    const float d = c.disp();
    if (white_bkgd) {                                                        // model_utils.py:97-98: rgb + (1 - acc)
        const float rest = __fsub_rn(1.f, c.acc);
        c.r = __fadd_rn(c.r, rest); c.g = __fadd_rn(c.g, rest); c.b = __fadd_rn(c.b, rest);
    }
    float* rgb = fine ? o.rgb : o.rgb_coarse;
    float* depth = fine ? o.depth : o.depth_coarse;
    float* acc = fine ? o.acc : o.acc_coarse;
    float* disp = fine ? o.disp : o.disp_coarse;
    if (rgb) { rgb[idx * 3] = c.r; rgb[idx * 3 + 1] = c.g; rgb[idx * 3 + 2] = c.b; }
    if (depth) depth[idx] = c.depth;
    if (acc) acc[idx] = c.acc;
    if (disp) disp[idx] = d;
    const uint32_t f = ((bad(c.r) || bad(c.g) || bad(c.b)) ? NWE_FLAG_RGB : 0) | (bad(c.depth) ? NWE_FLAG_DEPTH : 0) |
                       (bad(c.acc) ? NWE_FLAG_ACC : 0) | (bad(d) ? NWE_FLAG_DISP : 0);
    return fine ? f : (f << 4);   // the coarse flag bits sit 4 above the fine ones
}

__device__ __forceinline__ bool wants_survey(const nwe_outputs& o) {
    return o.z_std || o.sample_cond || o.sample_amp || o.sample_switch;
}
__device__ __forceinline__ uint32_t store_survey(const nwe_outputs& o, int64_t idx, const SampleSurvey& sv) {
    if (o.z_std) o.z_std[idx] = sv.z_std;
    if (o.sample_cond) o.sample_cond[idx] = sv.min_denom;
    if (o.sample_amp) o.sample_amp[idx] = sv.max_amp;
    if (o.sample_switch) o.sample_switch[idx] = sv.min_switch;
    return (o.z_std && bad(sv.z_std)) ? NWE_FLAG_ZSTD : 0;
}

}  // namespace nwe
