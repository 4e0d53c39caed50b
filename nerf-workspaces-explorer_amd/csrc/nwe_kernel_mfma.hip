// MFMA render kernel (NWE_PREC_F16X3 / NWE_PREC_F16X1) for gfx950.
//
// The whole path of nerf/inference/nerf_replica_inference_handler.py:203-277 in one launch: rays,
// coarse depths, gamma(x)/gamma(d), coarse MLP, compositing, inverse-CDF importance sampling + merge,
// fine MLP, compositing.  Nothing but the per-ray results reaches HBM.
//
// Work decomposition:
//   * one wavefront owns a packet of 32 rays (lane&31 = ray, both lane halves carry the ray state) and
//     walks their samples in lock step; a 256-thread workgroup = 4 packets sharing one weight stream.
//   * the MLP is evaluated transposed, H_out^T[feature, ray] = W[feature, k] . H_in^T[k, ray], with
//     v_mfma_f32_32x32x16_f16: A = weight tile (from LDS), B = activations.  The 32x32 result has the
//     ray on the lane and the features in the 16 registers, which is exactly the B-operand layout of
//     the next layer (k order permuted; the packer permutes the weight columns to match), so
//     activations never leave the register file between layers.
//   * fp32-grade results from fp16 MFMA: every operand is split x = hi + lo*2^-11 (both fp16) and
//     W.x ~= Whi.xhi + 2^-11 (Wlo.xhi + Whi.xlo); the 2^-11 terms get their own accumulator.
//   * weights stream from L2 through two LDS buffers with LDS-DMA (global_load_lds_dwordx4), one
//     chunk = one 32-row tile of a layer (bias tile + hi/lo tile per 16-wide k-step), prefetched one
//     chunk ahead.
#include "nwe_host.h"

namespace nwe {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int W>
struct Shape {
    static constexpr int NT = W / 32;    // 32-row tiles of a W-wide layer
    static constexpr int KH = W / 16;    // k-steps over a W-wide activation vector
    static constexpr int KG = 4;         // k-steps over gamma(x) (63 -> 64 slots)
    static constexpr int KD = 2;         // k-steps over gamma(d) (27 -> 32 slots)
    static constexpr int NTV = W / 64;   // row tiles of the view layer (W/2 outputs)
    static constexpr int KV = W / 32;    // k-steps over the view layer output
    static constexpr int T_L0 = 1 + 2 * KG;          // tiles per chunk: bias tile + (hi, lo) per k-step
    static constexpr int T_H = 1 + 2 * KH;
    static constexpr int T_S = 1 + 2 * (KG + KH);    // skip layer
    static constexpr int T_V = 1 + 2 * (KH + KD);
    static constexpr int T_RGB = 1 + 2 * KV;
    static constexpr int CHUNK_BYTES = T_S * kTileBytes;
};

constexpr int kWaves = 4;
constexpr int kRaysPerWave = 32;
constexpr float kLoScale = 2048.f;        // 2^kLoShift
constexpr float kLoInv = 1.f / 2048.f;

// Split 16 fp32 values (one 32x32 accumulator tile column) into the B fragments of two k-steps.
// value -> hi = fp16(v), lo = fp16((v - hi) * 2^11).  Register r of the tile is element r&7 of k-step r>>3.
template <bool X3>
__device__ __forceinline__ void split_tile(const f16v& v, h8& hi0, h8& lo0, h8& hi1, h8& lo1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const _Float16 h = (_Float16)v[r];
        _Float16 l = (_Float16)0.f;
        if (X3) l = (_Float16)((v[r] - (float)h) * kLoScale);
        if (r < 8) { hi0[r] = h; lo0[r] = l; } else { hi1[r - 8] = h; lo1[r - 8] = l; }
    }
}

template <bool X3>
__device__ __forceinline__ void mma3(const h8& a_hi, const h8& a_lo, const h8& x_hi, const h8& x_lo, f16v& acc1,
                                     f16v& acc2) {
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_hi, acc1, 0, 0, 0);
    if (X3) {
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, x_hi, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_lo, acc2, 0, 0, 0);
    }
}

// The weight stream of one network, walked chunk by chunk through two LDS buffers.
template <int CHUNK_BYTES>
struct Walker {
    const uint8_t* stream;
    uint32_t next_tile;   // first tile of the next chunk to issue
    char* buf0;
    int parity;           // buffer the next issue writes
    int wave, lane;

    __device__ __forceinline__ void start(const uint8_t* s) { stream = s; next_tile = 0; }
    __device__ __forceinline__ void issue(int ntiles) {
        char* dst = buf0 + parity * CHUNK_BYTES;
        const uint8_t* src = stream + (size_t)next_tile * kTileBytes + lane * 16;
        for (int t = wave; t < ntiles; t += kWaves) {
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + (size_t)t * kTileBytes),
                                             (LDS_AS void*)(dst + t * kTileBytes), 16, 0, 0);
        }
        next_tile += ntiles;
        parity ^= 1;
    }
    // Make the chunk issued last visible to every wave, start the next one, return the visible chunk.
    __device__ __forceinline__ const char* advance(int next_ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* cur = buf0 + (parity ^ 1) * CHUNK_BYTES;
        if (next_ntiles > 0) issue(next_ntiles);
        return cur;
    }
};

// One 32-row tile: acc = bias + sum over the chunk's k-steps.  Chunk = [bias tile][optional NKG gamma
// k-steps][NKH hidden k-steps], each k-step = hi tile then lo tile, each tile lane-linear (16 B/lane).
template <int NKG, int NKH, bool X3>
__device__ __forceinline__ void tile_mma(const char* chunk, int lane, bool use_g, const h8* Ghi, const h8* Glo,
                                         const h8* Xhi, const h8* Xlo, f16v& acc1, f16v& acc2) {
    const float4* bp = reinterpret_cast<const float4*>(chunk);
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {   // register 4g+i holds row 8g + 4h + i
        const float4 b = bp[2 * g + h];
        acc1[4 * g + 0] = b.x; acc1[4 * g + 1] = b.y; acc1[4 * g + 2] = b.z; acc1[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    const char* p = chunk + kTileBytes + lane * 16;
    if (NKG > 0) {
        if (use_g) {
#pragma unroll
            for (int s = 0; s < NKG; ++s) {
                const h8 a_hi = *reinterpret_cast<const h8*>(p + (2 * s) * kTileBytes);
                const h8 a_lo = *reinterpret_cast<const h8*>(p + (2 * s + 1) * kTileBytes);
                mma3<X3>(a_hi, a_lo, Ghi[s], Glo[s], acc1, acc2);
            }
            p += NKG * 2 * kTileBytes;
        }
    }
#pragma unroll
    for (int s = 0; s < NKH; ++s) {
        const h8 a_hi = *reinterpret_cast<const h8*>(p + (2 * s) * kTileBytes);
        const h8 a_lo = *reinterpret_cast<const h8*>(p + (2 * s + 1) * kTileBytes);
        mma3<X3>(a_hi, a_lo, Xhi[s], Xlo[s], acc1, acc2);
    }
}

template <bool X3>
__device__ __forceinline__ f16v finish(const f16v& acc1, const f16v& acc2, float lower) {
    f16v v;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float x = X3 ? __builtin_fmaf(acc2[r], kLoInv, acc1[r]) : acc1[r];
        v[r] = fmaxf(x, lower);
    }
    return v;
}

// A full layer: NT row tiles, input = [optional gamma k-steps] + NKH k-steps of X, output split into Y.
template <int NT, int NKG, int NKH, bool X3, class WalkerT>
__device__ __forceinline__ void layer(WalkerT& wk, int lane, bool use_g, int tiles_this, int tiles_after, const h8* Ghi,
                                      const h8* Glo, const h8* Xhi, const h8* Xlo, h8* Yhi, h8* Ylo, float lower) {
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        const char* chunk = wk.advance(rt + 1 < NT ? tiles_this : tiles_after);
        f16v acc1, acc2;
        tile_mma<NKG, NKH, X3>(chunk, lane, use_g, Ghi, Glo, Xhi, Xlo, acc1, acc2);
        const f16v v = finish<X3>(acc1, acc2, lower);
        split_tile<X3>(v, Yhi[2 * rt], Ylo[2 * rt], Yhi[2 * rt + 1], Ylo[2 * rt + 1]);
    }
}

// gamma(x) and gamma(d) slot maps (must match the packer, nwe_abi.hip: gamma_col()):
//   lane half h computes bands [NB*h, NB*h + NB) for the three coordinates; slot q = 2*pair + {0: sin, 1: cos},
//   pair = band_local*3 + coord; after the 6*NB sin/cos slots: identity slots (h=0: x, y; h=1: z, pad).
template <int NB, int NK, bool X3>
__device__ __forceinline__ void encode(float vx, float vy, float vz, int h, h8* Ehi, h8* Elo) {
    float vals[NK * 8];
#pragma unroll
    for (int i = 0; i < NK * 8; ++i) vals[i] = 0.f;
    const float first = h ? (float)(1 << NB) : 1.f;   // 2^(NB*h): exact scaling
#pragma unroll
    for (int bl = 0; bl < NB; ++bl) {
        const float f = first * (float)(1 << bl);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = c == 0 ? vx : (c == 1 ? vy : vz);
            float sn, cs;
            sincosf(v * f, &sn, &cs);                    // embedding.py:36: fn(x * freq), x*freq exact
            vals[2 * (bl * 3 + c)] = sn;
            vals[2 * (bl * 3 + c) + 1] = cs;
        }
    }
    vals[6 * NB] = h ? vz : vx;
    vals[6 * NB + 1] = h ? 0.f : vy;
#pragma unroll
    for (int s = 0; s < NK; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = vals[s * 8 + j];
            const _Float16 hh = (_Float16)v;
            Ehi[s][j] = hh;
            Elo[s][j] = X3 ? (_Float16)((v - (float)hh) * kLoScale) : (_Float16)0.f;
        }
    }
}

// One MLP evaluation for the wave's 32 points.  nerf/models/nerf_model.py:45-83.
// Trunk layers 1..D-1 and the feature layer run as (D/2) pairs A->B, B->A so that the two activation
// register sets keep fixed names inside a rolled loop.
template <int W, int D, int SKIP, bool X3, class WalkerT>
__device__ __forceinline__ void mlp_eval(WalkerT& wk, int lane, const h8* Ghi, const h8* Glo, const h8* GDhi,
                                         const h8* GDlo, float& o_r, float& o_g, float& o_b, float& o_s) {
    using S = Shape<W>;
    static_assert(D % 2 == 0, "trunk depth must be even");
    static_assert(SKIP < 0 || SKIP % 2 == 0, "skip layer index must be even");
    h8 Ahi[S::KH], Alo[S::KH], Bhi[S::KH], Blo[S::KH];
    constexpr int NPAIR = D / 2;
    constexpr int SKIP_PAIR = SKIP < 0 ? -1 : SKIP / 2;   // pair whose first layer takes [gamma, h]
    auto first_tiles = [&](int pair) { return pair == SKIP_PAIR ? S::T_S : S::T_H; };

    // layer 0: gamma(x) -> A
    layer<S::NT, 0, S::KG, X3>(wk, lane, false, S::T_L0, first_tiles(0), nullptr, nullptr, Ghi, Glo, Ahi, Alo, 0.f);

    float sigma = 0.f;
#pragma unroll 1
    for (int pair = 0; pair < NPAIR; ++pair) {
        const bool use_g = pair == SKIP_PAIR;
        const bool last = pair == NPAIR - 1;
        // first of pair: A (+gamma) -> B, ReLU
        layer<S::NT, S::KG, S::KH, X3>(wk, lane, use_g, first_tiles(pair), S::T_H, Ghi, Glo, Ahi, Alo, Bhi, Blo, 0.f);
        // second of pair: B -> A; the last pair's second layer is _feature_linear (no ReLU, nerf_model.py:64)
        const int after = last ? S::T_H /* alpha tile */ : first_tiles(pair + 1);
        layer<S::NT, 0, S::KH, X3>(wk, lane, false, S::T_H, after, nullptr, nullptr, Bhi, Blo, Ahi, Alo,
                                   last ? -INFINITY : 0.f);
        if (last) {   // _alpha_linear on the same input B (nerf_model.py:63); rows 0 and 4 of its tile both hold it
            const char* chunk = wk.advance(S::T_V);
            f16v acc1, acc2;
            tile_mma<0, S::KH, X3>(chunk, lane, false, nullptr, nullptr, Bhi, Blo, acc1, acc2);
            sigma = X3 ? __builtin_fmaf(acc2[0], kLoInv, acc1[0]) : acc1[0];
        }
    }
    // view layer: [feature (A), gamma(d)] -> B[0..KV), ReLU (nerf_model.py:66-70)
#pragma unroll
    for (int rt = 0; rt < S::NTV; ++rt) {
        const char* chunk = wk.advance(rt + 1 < S::NTV ? S::T_V : S::T_RGB);
        f16v acc1, acc2;
        tile_mma<0, S::KH, X3>(chunk, lane, false, nullptr, nullptr, Ahi, Alo, acc1, acc2);
        // gamma(d) k-steps follow the hidden ones in the chunk
        const char* p = chunk + (1 + 2 * S::KH) * kTileBytes + lane * 16;
#pragma unroll
        for (int s = 0; s < S::KD; ++s) {
            const h8 a_hi = *reinterpret_cast<const h8*>(p + (2 * s) * kTileBytes);
            const h8 a_lo = *reinterpret_cast<const h8*>(p + (2 * s + 1) * kTileBytes);
            mma3<X3>(a_hi, a_lo, GDhi[s], GDlo[s], acc1, acc2);
        }
        const f16v v = finish<X3>(acc1, acc2, 0.f);
        split_tile<X3>(v, Bhi[2 * rt], Blo[2 * rt], Bhi[2 * rt + 1], Blo[2 * rt + 1]);
    }
    // rgb head (nerf_model.py:74): rows 0..2 (and their copies 4..6 for the upper lane half)
    {
        const char* chunk = wk.advance(0);
        f16v acc1, acc2;
        tile_mma<0, S::KV, X3>(chunk, lane, false, nullptr, nullptr, Bhi, Blo, acc1, acc2);
        o_r = X3 ? __builtin_fmaf(acc2[0], kLoInv, acc1[0]) : acc1[0];
        o_g = X3 ? __builtin_fmaf(acc2[1], kLoInv, acc1[1]) : acc1[1];
        o_b = X3 ? __builtin_fmaf(acc2[2], kLoInv, acc1[2]) : acc1[2];
    }
    o_s = sigma;
}

template <int W>
struct Smem {
    static constexpr int CHUNKS = 2 * Shape<W>::CHUNK_BYTES;
    static constexpr int WOFF = CHUNKS;                                          // per-wave coarse weights / cdf
    static constexpr int TOFF = WOFF + kWaves * kMaxSamples * kRaysPerWave * 4;  // t, 1-t, u tables
    static constexpr int TOTAL = TOFF + (2 * kMaxSamples + kMaxImportance) * 4;
};

template <int W, int D, int SKIP, bool X3>
__global__ void __launch_bounds__(256) render_mfma_kernel(RenderArgs a, NetMfma nc, NetMfma nf) {
    using S = Shape<W>;
    using SM = Smem<W>;
    __shared__ __attribute__((aligned(16))) char smem[SM::TOTAL];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int ns = a.n_samples, ni = a.n_importance;

    float* s_t = reinterpret_cast<float*>(smem + SM::TOFF);
    float* s_omt = s_t + kMaxSamples;
    float* s_u = s_omt + kMaxSamples;
    for (int i = threadIdx.x; i < ns; i += 256) { s_t[i] = a.t_vals[i]; s_omt[i] = a.omt_vals[i]; }
    for (int i = threadIdx.x; i < ni; i += 256) s_u[i] = a.u_vals[i];

    const int64_t ridx = ((int64_t)blockIdx.x * kWaves + wave) * kRaysPerWave + (lane & 31);
    const bool live = ridx < a.n_rays && half == 0;
    const int64_t rclamp = ridx < a.n_rays ? ridx : a.n_rays - 1;
    const Ray ray = load_ray(a, rclamp);

    Walker<S::CHUNK_BYTES> wk;
    wk.buf0 = smem; wk.parity = 0; wk.wave = wave; wk.lane = lane;

    // gamma(d): once per ray (model_utils.py:23-25 re-embeds the same direction for every sample)
    h8 GDhi[S::KD], GDlo[S::KD];
    encode<2, S::KD, X3>(ray.vx, ray.vy, ray.vz, half, GDhi, GDlo);

    FineSampler fs;
    fs.wc = reinterpret_cast<float*>(smem + SM::WOFF) + wave * (kMaxSamples * kRaysPerWave) + (lane & 31);
    fs.stride = kRaysPerWave; fs.t_tab = s_t; fs.omt_tab = s_omt; fs.u_tab = s_u; fs.ns = ns; fs.ni = ni;
    __syncthreads();

    Composite comp;
    uint32_t flags = 0;
    for (int pass = 0; pass < (ni > 0 ? 2 : 1); ++pass) {
        const NetMfma& net = pass == 0 ? nc : nf;
        const int Stot = pass == 0 ? ns : ns + ni;
        comp.reset();
        float z_cur, z_next = 0.f;
        if (pass == 0) z_cur = coarse_z(ray, s_t[0], s_omt[0]);
        else {
            fs.prepare(ray);
            z_cur = a.z_fine_in ? a.z_fine_in[rclamp * Stot] : fs.next(ray);
        }
        for (int s = 0; s < Stot; ++s) {
            wk.start(net.stream);
            wk.issue(S::T_L0);   // first chunk of this evaluation flies while gamma(x) is computed
            if (s + 1 < Stot) {
                if (pass == 0) z_next = coarse_z(ray, s_t[s + 1], s_omt[s + 1]);
                else z_next = a.z_fine_in ? a.z_fine_in[rclamp * Stot + s + 1] : fs.next(ray);
            }
            float px, py, pz;
            point_at(ray, z_cur, px, py, pz);
            h8 Ghi[S::KG], Glo[S::KG];
            // handler.py:93: scalar_factor = 10, a true division (embedding.py:48)
            encode<5, S::KG, X3>(__fdiv_rn(px, 10.f), __fdiv_rn(py, 10.f), __fdiv_rn(pz, 10.f), half, Ghi, Glo);
            float rr, rg, rb, rs;
            mlp_eval<W, D, SKIP, X3>(wk, lane, Ghi, Glo, GDhi, GDlo, rr, rg, rb, rs);
            const float w = comp.step(rr, rg, rb, rs, z_cur, z_next, s + 1 == Stot, ray.dnorm);
            if (pass == 0) fs.wc[s * kRaysPerWave] = w;
            if (live) {
                float* raw = pass == 0 ? a.out.raw_coarse : a.out.raw_fine;
                if (raw) {
                    *reinterpret_cast<float4*>(raw + (ridx * Stot + s) * 4) = make_float4(rr, rg, rb, rs);
                    if (bad(rr) || bad(rg) || bad(rb) || bad(rs)) flags |= NWE_FLAG_RAW;
                }
                if (pass == 1 && a.out.z_fine) a.out.z_fine[ridx * Stot + s] = z_cur;
            }
            z_cur = z_next;
        }
        if (live) {
            flags |= store_ray(a.out, ridx, comp, pass == 1);
            if (ni == 0) flags |= store_ray(a.out, ridx, comp, true);
            if (pass == 1 && a.out.z_std) {
                const float zs = fs.z_std();
                a.out.z_std[ridx] = zs;
                if (bad(zs)) flags |= NWE_FLAG_ZSTD;
            }
            if (pass == 1 && a.out.sample_cond) a.out.sample_cond[ridx] = fs.min_denom;
        }
    }
    if (flags && a.out.flags) atomicOr(a.out.flags, flags);
}

bool mfma_supported(int D, int W, int in_xyz, int in_dir, int skip) {
    if (in_xyz != 63 || in_dir != 27) return false;
    return (D == 8 && W == 256 && skip == 4) || (D == 4 && W == 128 && skip == -1);
}

template <int W, int D, int SKIP>
static void launch_t(const RenderArgs& a, const NetMfma& nc, const NetMfma& nf, bool three_pass, hipStream_t stream) {
    const int64_t rays_per_block = kWaves * kRaysPerWave;
    const unsigned blocks = (unsigned)((a.n_rays + rays_per_block - 1) / rays_per_block);
    if (three_pass)
        hipLaunchKernelGGL((render_mfma_kernel<W, D, SKIP, true>), dim3(blocks), dim3(256), 0, stream, a, nc, nf);
    else
        hipLaunchKernelGGL((render_mfma_kernel<W, D, SKIP, false>), dim3(blocks), dim3(256), 0, stream, a, nc, nf);
}

bool launch_render_mfma(const RenderArgs& a, const NetMfma& nc, const NetMfma& nf, bool three_pass, hipStream_t stream) {
    const NetMfma& any = nc;
    if (a.n_importance > 0 && (nf.D != nc.D || nf.W != nc.W || nf.skip != nc.skip)) return false;
    if (any.D == 8 && any.W == 256 && any.skip == 4) launch_t<256, 8, 4>(a, nc, nf, three_pass, stream);
    else if (any.D == 4 && any.W == 128 && any.skip == -1) launch_t<128, 4, -1>(a, nc, nf, three_pass, stream);
    else return false;
    return true;
}

}  // namespace nwe
