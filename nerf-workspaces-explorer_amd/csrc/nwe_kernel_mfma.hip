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
//   * fp32-grade results from fp16 MFMA: every operand is split x = hi + lo (both fp16) and
//     W.x ~= Whi.xhi + Wlo.xhi + Whi.xlo, three MFMAs into one fp32 accumulator.  Weights are scaled
//     by a power of two at pack time so that their lo halves are fp16-normal; activation lo halves may
//     be fp16-subnormal (absolute error <= 3e-8), which the matrix core honours (nwe_selftest).
//   * weights stream from L2 through two LDS buffers with LDS-DMA (global_load_lds_dwordx4), one chunk
//     = one 32-row tile of a layer (hi/lo tile per 16-wide k-step), issued one chunk ahead, piece by
//     piece between the MFMAs of the current tile.
//   * one wave per SIMD: the wave's own instruction issue is the scarce resource next to the matrix
//     pipe, so everything around the MFMAs is kept to a handful of instructions per MFMA: the
//     epilogue of tile t (bias, ReLU, hi/lo split) is interleaved with the MFMAs of tile t+1, LDS-DMA
//     addressing is scalar, the A fragments are read three k-steps ahead.
#include "nwe_host.h"

namespace nwe {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define LDS_AS __attribute__((address_space(3)))

constexpr int kWaves = 4;
constexpr int kRaysPerWave = 32;

template <int W, int D>
struct Shape {
    static constexpr int NT = W / 32;    // 32-row tiles of a W-wide layer
    static constexpr int KH = W / 16;    // k-steps over a W-wide activation vector
    static constexpr int KG = 4;         // k-steps over gamma(x) (63 -> 64 slots)
    static constexpr int KD = 2;         // k-steps over gamma(d) (27 -> 32 slots)
    static constexpr int NTV = W / 64;   // row tiles of the view layer (W/2 outputs)
    static constexpr int KV = W / 32;    // k-steps over the view layer output
    // LDS-DMA pieces (1 KiB tiles) per wave and chunk: (hi, lo) per k-step, split evenly over the 4 waves
    static constexpr int N_L0 = 2 * KG / kWaves;
    static constexpr int N_H = 2 * KH / kWaves;
    static constexpr int N_S = 2 * (KH + KG) / kWaves;   // skip layer
    static constexpr int N_V = 2 * (KH + KD) / kWaves;
    static constexpr int N_RGB = 2 * KV / kWaves;
    static constexpr int CHUNK_BYTES = N_S * kWaves * kTileBytes;
    static constexpr int N_CHUNKS = NT + D * NT + 1 + NTV + 1;   // layer 0, D/2 pairs, alpha, views, rgb
};

// A 32-row tile whose accumulator is complete but whose epilogue (scale, bias, ReLU, fp16 hi/lo split into the B
// fragments of the next layer) has not run yet.  The epilogue of tile t is issued piecewise BETWEEN the MFMAs of
// tile t+1 (a wave issues in order: VALU placed between two MFMAs executes while the matrix pipe works), so two
// of these alternate.  The bias is read when the tile starts and consumed one tile later, which also keeps its
// LDS latency off the MFMA chain.
struct Pend {
    f16v a;
    float4 bias[4];   // register 4g+i holds row 8g + 4h + i -> bias[g].{x,y,z,w}
};

__device__ __forceinline__ float pend_value(const Pend& t, int r, float inv_scale) {
    const float4 b = t.bias[r >> 2];
    const float bb = (r & 3) == 0 ? b.x : ((r & 3) == 1 ? b.y : ((r & 3) == 2 ? b.z : b.w));
    return __builtin_fmaf(t.a[r], inv_scale, bb);
}

// Epilogue of elements 2p, 2p+1 of a pending tile: v = max(acc/scale + bias, lower), hi = fp16(v), lo = fp16(v - hi).
// Register r of the tile is element r&7 of the (r>>3)-th of its two output k-steps.
template <bool X3>
__device__ __forceinline__ void finish_pair(const Pend& t, int p, float inv_scale, float lower, h8& hi0, h8& lo0, h8& hi1,
                                            h8& lo1) {
#pragma unroll
    for (int e = 2 * p; e < 2 * p + 2; ++e) {
        const float v = fmaxf(pend_value(t, e, inv_scale), lower);
        const _Float16 h = (_Float16)v;
        const _Float16 l = X3 ? (_Float16)__builtin_fmaf((float)h, -1.f, v) : (_Float16)0.f;   // v_fma_mix: no separate cvt
        if (e < 8) { hi0[e] = h; lo0[e] = l; } else { hi1[e - 8] = h; lo1[e - 8] = l; }
    }
}

template <bool X3>
__device__ __forceinline__ void mma3(const h8& a_hi, const h8& a_lo, const h8& x_hi, const h8& x_lo, f16v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_hi, acc, 0, 0, 0);
    if (X3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, x_hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, x_lo, acc, 0, 0, 0);
    }
}

// The weight stream of one network, walked chunk by chunk through two LDS buffers.
//
// LDS-DMA goes through inline asm: hipcc's waitcnt pass treats a builtin global_load_lds as an LDS store that may
// alias every later ds_read of the same array and drains vmcnt(0) in front of the first one, which would serialise
// the prefetch with the compute it is meant to hide behind.  The asm form is invisible to that pass; completion is
// waited for by hand in advance() (s_waitcnt vmcnt(0) + barrier).  M0 carries the wave-uniform LDS destination and
// is compiler-reserved, so it is saved and restored inside the statement.  Wave w streams the w-th quarter of a
// chunk (n consecutive 1-KiB pieces): source = scalar base + lane*16, so a piece costs scalar instructions only.
template <int CHUNK_BYTES>
struct Walker {
    const uint8_t* stream;
    uint32_t next_tile;      // first tile of the next chunk to stream
    uint32_t lds_chunks;     // LDS byte address of chunk buffer 0
    const char* buf0;
    const float* bias_tab;   // LDS bias table of the current network, 32 floats per chunk
    int chunk;               // index of the chunk being consumed
    int parity;              // buffer the next chunk is written to
    int wave, dbg;
    uint32_t lane_off;       // lane * 16
    const uint8_t* blk_src;  // this wave's quarter of the chunk being streamed (uniform)
    uint32_t blk_dst;

    __device__ __forceinline__ void start(const uint8_t* s, const float* bias) {
        stream = s; bias_tab = bias; next_tile = 0; chunk = -1;
    }
    __device__ __forceinline__ void begin(int n_per_wave) {
        blk_src = stream + ((size_t)next_tile + (size_t)wave * n_per_wave) * kTileBytes;
        blk_dst = lds_chunks + parity * CHUNK_BYTES + wave * n_per_wave * kTileBytes;
        next_tile += n_per_wave * kWaves;
        parity ^= 1;
    }
    __device__ __forceinline__ void piece(int i) {
        if (dbg & 1) return;
        const uint8_t* src = blk_src + (size_t)i * kTileBytes;
        const uint32_t dst = blk_dst + i * kTileBytes;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(src), "s"(dst) : "memory");
    }
    // Make the chunk streamed last visible to every wave and return it.
    __device__ __forceinline__ const char* advance() {
        if (!(dbg & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        ++chunk;
        return buf0 + (parity ^ 1) * CHUNK_BYTES;
    }
};

// NK k-steps of one tile with the A fragments (hi, lo tile pairs at p, lane-linear) read PD k-steps ahead of their
// MFMAs and, if PEND, the epilogue of the previous tile spread over the k-steps (pairs [8s/NK, 8(s+1)/NK) are
// finished BEFORE the MFMAs of k-step s, so a pending tile that feeds this tile's last two k-steps is complete in
// time).  FIRST: the accumulator starts from the literal zero of the first MFMA.  NPC > 0: this wave's NPC (+2 if
// `extra`) pieces of the next chunk are issued spread over the k-steps.
template <int NK, bool X3, bool PEND, bool FIRST, int NPC, class WalkerT>
__device__ __forceinline__ void ksteps(WalkerT& wk, const char* p, const h8* Xhi, const h8* Xlo, f16v& acc, const Pend& prev,
                                       float inv_scale, float lower, bool extra, h8& y0h, h8& y0l, h8& y1h, h8& y1l) {
    constexpr int PD = 3;
    h8 fh[PD + 1], fl[PD + 1];
#pragma unroll
    for (int s = 0; s < PD && s < NK; ++s) {
        fh[s] = *reinterpret_cast<const h8*>(p + (2 * s) * kTileBytes);
        if (X3) fl[s] = *reinterpret_cast<const h8*>(p + (2 * s + 1) * kTileBytes);
    }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        if (NPC > 0) {
#pragma unroll
            for (int i = (s * NPC + NK - 1) / NK; i < ((s + 1) * NPC + NK - 1) / NK; ++i) wk.piece(i);
            if (s == NK - 1 && extra) { wk.piece(NPC); wk.piece(NPC + 1); }
        }
        if (s + PD < NK) {
            fh[(s + PD) % (PD + 1)] = *reinterpret_cast<const h8*>(p + (2 * (s + PD)) * kTileBytes);
            if (X3) fl[(s + PD) % (PD + 1)] = *reinterpret_cast<const h8*>(p + (2 * (s + PD) + 1) * kTileBytes);
        }
        if (PEND) {
#pragma unroll
            for (int q = (8 * s) / NK; q < (8 * (s + 1)) / NK; ++q) finish_pair<X3>(prev, q, inv_scale, lower, y0h, y0l, y1h, y1l);
        }
        if (FIRST && s == 0) {
            f16v zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[0], Xhi[0], zero, 0, 0, 0);
            if (X3) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[0], Xhi[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[0], Xlo[0], acc, 0, 0, 0);
            }
        } else {
            mma3<X3>(fh[s % (PD + 1)], fl[s % (PD + 1)], Xhi[s], Xlo[s], acc);
        }
        // Pin the issue order of this k-step (LLVM SchedGroupMask: 0x100 DS read, 0x8 MFMA, 0x2 VALU): the fragment
        // reads of k-step s+PD, then each MFMA followed by a few epilogue VALU ops that execute while the matrix pipe
        // works.  Without this hipcc sinks the reads next to their use and clusters the epilogue.
        if (s + PD < NK) __builtin_amdgcn_sched_group_barrier(0x100, X3 ? 2 : 1, 0);
        constexpr int V = PEND ? ((NK >= 16) ? (X3 ? 2 : 6) : ((NK >= 8) ? (X3 ? 4 : 12) : 8)) : 0;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (V) __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
        if (X3) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (V) __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (V) __builtin_amdgcn_sched_group_barrier(0x002, V, 0);
        }
    }
    // Keep the epilogue HERE: its results are only consumed by the next layer, so without a use at this point
    // LLVM sinks the whole epilogue of every tile of a layer to the layer's end (and keeps all their accumulators
    // alive), which is exactly the un-overlapped VALU block this structure is meant to remove.
    if (PEND) asm volatile("" : "+v"(y0h), "+v"(y0l), "+v"(y1h), "+v"(y1l));
}

// One 32-row tile.  Chunk = [NKH k-steps over X][NKG gamma k-steps if use_g][NKD k-steps over D], each k-step =
// hi tile then lo tile, each tile lane-linear (16 B per lane).  The next chunk (NPC pieces per wave, +2 if `extra`)
// is streamed from inside the X segment.
template <int NKG, int NKH, int NKD, bool X3, bool PEND, int NPC, class WalkerT>
__device__ __forceinline__ void tile_mma(WalkerT& wk, int lane, bool use_g, bool extra, const h8* Ghi, const h8* Glo, const h8* Xhi,
                                         const h8* Xlo, const h8* Dhi, const h8* Dlo, Pend& cur, const Pend& prev, float inv_scale,
                                         float lower, h8& y0h, h8& y0l, h8& y1h, h8& y1l) {
    const char* chunk = wk.advance();
    if (NPC > 0) wk.begin(NPC + (extra ? 2 : 0));
    const float4* bp = reinterpret_cast<const float4*>(wk.bias_tab + wk.chunk * 32);
    const int h = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) cur.bias[g] = bp[2 * g + h];
    // the 4 bias reads and the fragment reads of the first three k-steps go first
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + (X3 ? 2 : 1) * (NKH < 3 ? NKH : 3), 0);
    const char* p = chunk + lane * 16;
    ksteps<NKH, X3, PEND, true, NPC>(wk, p, Xhi, Xlo, cur.a, prev, inv_scale, lower, extra, y0h, y0l, y1h, y1l);
    p += NKH * 2 * kTileBytes;
    if (NKG > 0) {
        if (use_g) {
            h8 d0, d1, d2, d3;
            ksteps<NKG, X3, false, false, 0>(wk, p, Ghi, Glo, cur.a, prev, inv_scale, lower, false, d0, d1, d2, d3);
        }
    }
    if (NKD > 0) {
        h8 d0, d1, d2, d3;
        ksteps<NKD, X3, false, false, 0>(wk, p, Dhi, Dlo, cur.a, prev, inv_scale, lower, false, d0, d1, d2, d3);
    }
}

// A full layer of NT tiles reading X (+ gamma k-steps) and writing Y.  Tile rt accumulates into P[rt&1] while the
// epilogue of the tile before it runs: for rt = 0 that is the LAST tile of the previous layer (in P1, destined for
// k-steps 2*NT-2, 2*NT-1 of X itself), for rt > 0 tile rt-1 of this layer (destined for Y).  On return P1 holds
// this layer's last tile, still pending.  The chunks of this layer have 2*NKH/4 pieces per wave (+2 with the gamma
// k-steps, i.e. when use_g); the chunk after the layer's last has NPC_AFTER (+2 if `extra_after`).
template <int NT, int NKG, int NKH, bool X3, bool PEND0, int NPC_AFTER, class WalkerT>
__device__ __forceinline__ void layer(WalkerT& wk, int lane, bool use_g, bool extra_after, const h8* Ghi, const h8* Glo, h8* Xhi,
                                      h8* Xlo, h8* Yhi, h8* Ylo, Pend& P0, Pend& P1, float inv_scale, float lower_prev, float lower) {
    static_assert(NT % 2 == 0, "tiles per layer must be even (accumulator ping-pong)");
    constexpr int NPC_THIS = 2 * NKH / kWaves;
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        Pend& cur = (rt & 1) ? P1 : P0;
        Pend& prev = (rt & 1) ? P0 : P1;
        if (rt == 0) {
            if constexpr (PEND0) {
                constexpr int L = 2 * NT - 2;   // the previous layer has as many tiles as X has k-step pairs
                tile_mma<NKG, NKH, 0, X3, true, NPC_THIS>(wk, lane, use_g, use_g, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur, prev,
                                                          inv_scale, lower_prev, Xhi[L], Xlo[L], Xhi[L + 1], Xlo[L + 1]);
            } else {
                h8 d0, d1, d2, d3;
                tile_mma<NKG, NKH, 0, X3, false, NPC_THIS>(wk, lane, use_g, use_g, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur, prev,
                                                           inv_scale, lower_prev, d0, d1, d2, d3);
            }
        } else if (rt + 1 < NT) {
            tile_mma<NKG, NKH, 0, X3, true, NPC_THIS>(wk, lane, use_g, use_g, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur, prev, inv_scale,
                                                      lower, Yhi[2 * rt - 2], Ylo[2 * rt - 2], Yhi[2 * rt - 1], Ylo[2 * rt - 1]);
        } else {
            tile_mma<NKG, NKH, 0, X3, true, NPC_AFTER>(wk, lane, use_g, extra_after, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur, prev,
                                                       inv_scale, lower, Yhi[2 * rt - 2], Ylo[2 * rt - 2], Yhi[2 * rt - 1], Ylo[2 * rt - 1]);
        }
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
            Elo[s][j] = X3 ? (_Float16)(v - (float)hh) : (_Float16)0.f;
        }
    }
}

// One MLP evaluation for the wave's 32 points.  nerf/models/nerf_model.py:45-83.
// Trunk layers 1..D-1 and the feature layer run as (D/2) pairs A->B, B->A so that the two activation register
// sets keep fixed names inside a rolled loop; every tile's epilogue is deferred into the next tile (see Pend).
// On entry the first chunk of the stream (layer 0, tile 0) is in flight.
template <int W, int D, int SKIP, bool X3, class WalkerT>
__device__ __forceinline__ void mlp_eval(WalkerT& wk, int lane, float inv_scale, h8* Ghi, h8* Glo, const h8* GDhi, const h8* GDlo,
                                         float& o_r, float& o_g, float& o_b, float& o_s) {
    using S = Shape<W, D>;
    static_assert(D % 2 == 0, "trunk depth must be even");
    static_assert(SKIP < 0 || SKIP % 2 == 0, "skip layer index must be even");
    static_assert(S::NT % 2 == 0 && S::NTV % 2 == 0, "tile counts must be even");
    h8 Ahi[S::KH], Alo[S::KH], Bhi[S::KH], Blo[S::KH];
    Pend P0, P1;
    constexpr int NPAIR = D / 2;
    constexpr int SKIP_PAIR = SKIP < 0 ? -1 : SKIP / 2;   // pair whose first layer takes [h, gamma]

    // layer 0: gamma(x) -> A (nothing pending in front of its first tile); the chunk after it opens pair 0
    layer<S::NT, 0, S::KG, X3, false, S::N_H>(wk, lane, false, SKIP_PAIR == 0, nullptr, nullptr, Ghi, Glo, Ahi, Alo, P0, P1, inv_scale,
                                              0.f, 0.f);
#pragma unroll 1
    for (int pair = 0; pair < NPAIR; ++pair) {
        const bool use_g = pair == SKIP_PAIR;
        const bool last = pair == NPAIR - 1;
        // first of pair: A (+gamma) -> B, ReLU.  Its first tile finishes the pending last tile of A (ReLU: the
        // producer is layer 0 or a non-final second-of-pair layer).
        layer<S::NT, S::KG, S::KH, X3, true, S::N_H>(wk, lane, use_g, false, Ghi, Glo, Ahi, Alo, Bhi, Blo, P0, P1, inv_scale, 0.f, 0.f);
        // second of pair: B -> A; the last pair's second layer is _feature_linear (no ReLU, nerf_model.py:64);
        // after it comes the next pair's first layer (skip: 2 more pieces) or the alpha tile
        layer<S::NT, 0, S::KH, X3, true, S::N_H>(wk, lane, false, !last && pair + 1 == SKIP_PAIR, nullptr, nullptr, Bhi, Blo, Ahi, Alo,
                                                 P0, P1, inv_scale, 0.f, last ? -INFINITY : 0.f);
    }
    constexpr int L = 2 * S::NT - 2;
    // _alpha_linear on B, the input of _feature_linear (nerf_model.py:63); meanwhile the last feature tile (P1) is
    // finished into A without ReLU.  Rows 0 and 4 of the alpha tile both hold the single output row.
    tile_mma<0, S::KH, 0, X3, true, S::N_V>(wk, lane, false, false, nullptr, nullptr, Bhi, Blo, nullptr, nullptr, P0, P1, inv_scale,
                                            -INFINITY, Ahi[L], Alo[L], Ahi[L + 1], Alo[L + 1]);
    const float sigma = pend_value(P0, 0, inv_scale);
    // view layer: [feature (A), gamma(d)] -> B[0..KV), ReLU (nerf_model.py:66-70); tile rt accumulates in P[(rt+1)&1]
#pragma unroll
    for (int rt = 0; rt < S::NTV; ++rt) {
        Pend& cur = (rt & 1) ? P0 : P1;
        Pend& prev = (rt & 1) ? P1 : P0;
        if (rt == 0) {
            h8 d0, d1, d2, d3;   // the alpha tile (P0) has no activation output
            tile_mma<0, S::KH, S::KD, X3, false, S::N_V>(wk, lane, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo, cur, prev,
                                                         inv_scale, 0.f, d0, d1, d2, d3);
        } else if (rt + 1 < S::NTV) {
            tile_mma<0, S::KH, S::KD, X3, true, S::N_V>(wk, lane, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo, cur, prev,
                                                        inv_scale, 0.f, Bhi[2 * rt - 2], Blo[2 * rt - 2], Bhi[2 * rt - 1], Blo[2 * rt - 1]);
        } else {
            tile_mma<0, S::KH, S::KD, X3, true, S::N_RGB>(wk, lane, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo, cur, prev,
                                                          inv_scale, 0.f, Bhi[2 * rt - 2], Blo[2 * rt - 2], Bhi[2 * rt - 1], Blo[2 * rt - 1]);
        }
    }
    // rgb head (nerf_model.py:74) in P1 while the last view tile (P0, NTV even) is finished into B; rows 0..2 and
    // their copies 4..6 for the upper lane half.  Nothing is streamed behind it: the caller starts the next pass.
    {
        constexpr int LV = 2 * S::NTV - 2;
        tile_mma<0, S::KV, 0, X3, true, 0>(wk, lane, false, false, nullptr, nullptr, Bhi, Blo, nullptr, nullptr, P1, P0, inv_scale, 0.f,
                                           Bhi[LV], Blo[LV], Bhi[LV + 1], Blo[LV + 1]);
    }
    o_r = pend_value(P1, 0, inv_scale);
    o_g = pend_value(P1, 1, inv_scale);
    o_b = pend_value(P1, 2, inv_scale);
    o_s = sigma;
}

constexpr int kMfmaMaxSamples = 64;   // coarse samples the per-wave LDS weight buffer is sized for

template <int W, int D>
struct Smem {
    using S = Shape<W, D>;
    static constexpr int CHUNKS = 2 * S::CHUNK_BYTES;
    static constexpr int BOFF = CHUNKS;                                              // bias tables, coarse then fine
    static constexpr int BIAS_BYTES = ((S::N_CHUNKS * 32 * 4 + 255) / 256) * 256;
    static constexpr int WOFF = BOFF + 2 * BIAS_BYTES;                               // per-wave coarse weights / cdf
    static constexpr int TOFF = WOFF + kWaves * kMfmaMaxSamples * kRaysPerWave * 4;  // t, 1-t, u tables
    static constexpr int TOTAL = TOFF + (2 * kMfmaMaxSamples + kMaxImportance) * 4;
    static_assert(TOTAL <= 160 * 1024, "LDS budget");
};

template <int W, int D, int SKIP, bool X3>
__global__ void __launch_bounds__(256) render_mfma_kernel(RenderArgs a, NetMfma nc, NetMfma nf) {
    using S = Shape<W, D>;
    using SM = Smem<W, D>;
    __shared__ __attribute__((aligned(16))) char smem[SM::TOTAL];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5;
    const int ns = a.n_samples, ni = a.n_importance;

    float* s_t = reinterpret_cast<float*>(smem + SM::TOFF);
    float* s_omt = s_t + kMfmaMaxSamples;
    float* s_u = s_omt + kMfmaMaxSamples;
    for (int i = threadIdx.x; i < ns; i += 256) { s_t[i] = a.t_vals[i]; s_omt[i] = a.omt_vals[i]; }
    for (int i = threadIdx.x; i < ni; i += 256) s_u[i] = a.u_vals[i];
    float* s_bias = reinterpret_cast<float*>(smem + SM::BOFF);
    for (int i = threadIdx.x; i < S::N_CHUNKS * 32; i += 256) {
        s_bias[i] = nc.bias[i];
        if (ni > 0) s_bias[SM::BIAS_BYTES / 4 + i] = nf.bias[i];
    }

    const int64_t ridx = ((int64_t)blockIdx.x * kWaves + wave) * kRaysPerWave + (lane & 31);
    const bool live = ridx < a.n_rays && half == 0;
    const int64_t rclamp = ridx < a.n_rays ? ridx : a.n_rays - 1;
    const Ray ray = load_ray(a, rclamp);

    Walker<S::CHUNK_BYTES> wk;
    wk.buf0 = smem; wk.lds_chunks = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    wk.parity = 0; wk.wave = wave; wk.lane_off = lane * 16; wk.dbg = a.dbg;

    // gamma(d): once per ray (model_utils.py:23-25 re-embeds the same direction for every sample)
    h8 GDhi[S::KD], GDlo[S::KD];
    encode<2, S::KD, X3>(ray.vx, ray.vy, ray.vz, half, GDhi, GDlo);

    FineSampler fs;
    fs.wc = reinterpret_cast<float*>(smem + SM::WOFF) + wave * (kMfmaMaxSamples * kRaysPerWave) + (lane & 31);
    fs.stride = kRaysPerWave; fs.t_tab = s_t; fs.omt_tab = s_omt; fs.u_tab = s_u; fs.ns = ns; fs.ni = ni;
    __syncthreads();

    Composite comp;
    uint32_t flags = 0;
    for (int pass = 0; pass < (ni > 0 ? 2 : 1); ++pass) {
        const NetMfma& net = pass == 0 ? nc : nf;
        const float* bias = s_bias + (pass == 0 ? 0 : SM::BIAS_BYTES / 4);
        const int Stot = pass == 0 ? ns : ns + ni;
        comp.reset();
        float z_cur, z_next = 0.f;
        if (pass == 0) z_cur = coarse_z(ray, s_t[0], s_omt[0]);
        else {
            fs.prepare(ray);
            z_cur = a.z_fine_in ? a.z_fine_in[rclamp * Stot] : fs.next(ray);
        }
        for (int s = 0; s < Stot; ++s) {
            wk.start(net.stream, bias);
            wk.begin(S::N_L0);   // first chunk of this evaluation flies while gamma(x) is computed
#pragma unroll
            for (int i = 0; i < S::N_L0; ++i) wk.piece(i);
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
            mlp_eval<W, D, SKIP, X3>(wk, lane, net.inv_scale, Ghi, Glo, GDhi, GDlo, rr, rg, rb, rs);
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

int mfma_max_samples() { return kMfmaMaxSamples; }

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
    if (a.n_importance > 0 && (nf.D != nc.D || nf.W != nc.W || nf.skip != nc.skip)) return false;
    if (a.n_samples > kMfmaMaxSamples) return false;
    if (nc.D == 8 && nc.W == 256 && nc.skip == 4) launch_t<256, 8, 4>(a, nc, nf, three_pass, stream);
    else if (nc.D == 4 && nc.W == 128 && nc.skip == -1) launch_t<128, 4, -1>(a, nc, nf, three_pass, stream);
    else return false;
    return true;
}

}  // namespace nwe
