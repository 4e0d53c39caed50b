// MFMA render kernel (NWE_PREC_F16X3 / NWE_PREC_F16X1) for gfx950.
//
// The whole path of nerf/inference/nerf_replica_inference_handler.py:203-277 in one launch: rays,
// coarse depths, gamma(x)/gamma(d), coarse MLP, compositing, inverse-CDF importance sampling + merge,
// fine MLP, compositing.  Nothing but the per-ray results reaches HBM.
//
// Work decomposition:
//   * a packet = 32 rays (lane&31 = ray, both lane halves carry the ray state) whose samples are walked in
//     lock step; a 256-thread workgroup = 4 waves sharing one weight stream, either four packets (one per
//     wave) or one packet with its samples dealt to the four waves (render_mfma_kernel, SPLIT).
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
//     pipe, so everything around the MFMAs is kept to a handful of instructions per MFMA and placed
//     statically in the gaps between them: the epilogue of tile t (bias, ReLU, hi/lo split) runs in
//     stages between the MFMAs of tile t+1 (EpiPlan), LDS-DMA addressing is scalar and each piece has a
//     gap of its own (DmaPlan), the A fragments are read three k-steps ahead.
//   * template switches of render_mfma_kernel: X3 (three split products / single fp16 product), SPLIT (one packet per
//     workgroup, samples dealt to the waves), FORM (nwe_host.h: kFormFolded = _feature_linear multiplied into the view layer by
//     the packer and _alpha_linear a dot product, the product path - "FOLD" below; kFormReference = every layer a tile of the
//     stream; kFormNoViewDirs = trunk + _output_linear), LEAN (only rgb / depth / acc of pinhole views: every other pointer
//     compile-time null, no register spills).
// This header holds the templates; nwe_mfma_inst_*.hip instantiate them (in parallel), nwe_kernel_mfma.hip dispatches.
#pragma once
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
    static constexpr int N_CHUNKS = NT + D * NT + 1 + NTV + 1;   // layer 0, D-1 trunk layers + feature, alpha, views, rgb (unfolded: the larger count)
    // FOLD (the product path): _feature_linear is folded into the view layer at pack time (-NT chunks) and _alpha_linear is not
    // a tile of the stream at all (-1): its single output row is a dot product with the last trunk layer's activations,
    // accumulated in fp32 on the vector ALU inside that layer's epilogue (see mlp_eval).  Its weights travel as NT extra
    // rows of the bias table (row rt, element i = weight of trunk feature 32 rt + i) plus one row whose element 0 is its bias.
    static constexpr int N_CHUNKS_FOLDED = N_CHUNKS - NT - 1;
    static constexpr int N_DOT_ROWS = NT + 1;
    // kFormNoViewDirs: layer 0, D-1 trunk layers, one chunk of _output_linear (nerf_model.py:42-43,78-79)
    static constexpr int N_CHUNKS_NOVIEW = D * NT + 1;
    static constexpr int n_chunks(int form) { return form == kFormFolded ? N_CHUNKS_FOLDED : (form == kFormNoViewDirs ? N_CHUNKS_NOVIEW : N_CHUNKS); }
    static constexpr int n_bias_rows(int form) { return n_chunks(form) + (form == kFormFolded ? N_DOT_ROWS : 0); }
    // A LONG chunk (>= 16 k-steps, three-pass mode) keeps the (hi, lo) tiles of its last k-step in a rotating tail slot
    // instead of the chunk buffer, see Walker.
    static constexpr int LONG_PIECES = 8;
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

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// Epilogue of a pending tile: v = max(acc/scale + bias, lower), hi = fp16(v), lo = fp16(v - hi); register r of the tile
// is element r&7 of the (r>>3)-th of its two output k-steps.  This is the FALLBACK form, one element per call (every
// second call packs a pair), used only by tiles too short for the staged plan below (EpiPlan::STAGED == false).
template <bool X3, bool DOT = false>
__device__ __forceinline__ void finish_elem(const Pend& t, int e, float inv_scale, float lower, float& keep, h8& hi0, h8& lo0,
                                            h8& hi1, h8& lo1, const float* dotw = nullptr, float* dot = nullptr) {
    const float v = fmaxf(pend_value(t, e, inv_scale), lower);
    if (DOT) *dot = __builtin_fmaf(dotw[8 * (e >> 2) + (e & 3)], v, *dot);   // dotw already points at this lane half's rows
    if ((e & 1) == 0) { keep = v; return; }
    const float v0 = keep, v1 = v;
    h2 hp;
    hp[0] = (_Float16)v0; hp[1] = (_Float16)v1;                        // one v_cvt_pk_f16_f32
    const _Float16 h0 = hp[0], h1 = hp[1];
    _Float16 l0 = (_Float16)0.f, l1 = (_Float16)0.f;
    if (X3) {
        // residual v - hi as fma(hi, -1, v) with the fp16 half read in place: v_fma_mix_f32 instead of v_cvt_f32_f16 +
        // v_sub_f32 (same single rounding).  hipcc does not select it from C, hence the asm.
        const uint32_t hw = __builtin_bit_cast(uint32_t, hp);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw), "v"(v1));
        l0 = (_Float16)r0; l1 = (_Float16)r1;
    }
    if (e < 8) { hi0[e - 1] = h0; hi0[e] = h1; lo0[e - 1] = l0; lo0[e] = l1; }
    else { hi1[e - 9] = h0; hi1[e - 8] = h1; lo1[e - 9] = l0; lo1[e - 8] = l1; }
}

// ---- staged epilogue ----------------------------------------------------------------------------------------------
// With one wave per SIMD an instruction costs 4 issue cycles, a VALU op that consumes the result of the instruction right
// in front of it 8, and about six independent ones hide behind one 32-cycle MFMA (tools/ubench/mfma_issue.hip).  The
// epilogue of a pending tile is therefore cut into STAGES of mutually independent ops over a group of elements, one stage
// per MFMA gap: read accumulators | fma | max | pack hi | residual | pack lo | park hi | park lo.  Groups follow each other
// from gap 1 on; a plan exists when all of it fits in front of the k-steps that consume the outputs.
struct Epi {
    float v[16];        // activation in fp32
    float r[16];        // residual v - hi
    uint32_t hp[8];     // packed fp16 pairs: hi
    uint32_t lp[8];     // lo
    float dw[16];       // DOT tiles: the dot-product weights of the elements (read one stage ahead of their use)
};

// PM: bit q set = k-step q issues a DMA piece.  A piece costs ~16 issue cycles (it is priced like a four-dword store),
// so in the three-pass kernel it has the gap behind the k-step's second MFMA to itself.
template <bool X3, int NKH, int NQ, bool FEEDS, uint32_t PM>
struct EpiPlan {
    static constexpr int GPK = X3 ? 3 : 1;                 // MFMA gaps per k-step
    static constexpr int NS = X3 ? 8 : 5;                  // stages per group
    static constexpr int NGAPS = GPK * NQ;
    static constexpr int D0 = FEEDS ? GPK * (NKH - 2) - 1 : NGAPS - 1;   // last gap for outputs 0..7
    static constexpr int D1 = FEEDS ? GPK * (NKH - 1) - 1 : NGAPS - 1;   // ... 8..15
    static constexpr bool usable(int gi) { return gi >= 1 && gi < NGAPS && !(X3 && gi % GPK == 1 && ((PM >> (gi / GPK)) & 1u)); }
    static constexpr int gap_of(int n) {                    // gap of the n-th stage slot
        int c = -1;
        for (int gi = 0; gi < NGAPS; ++gi)
            if (usable(gi) && ++c == n) return gi;
        return 1 << 20;
    }
    static constexpr bool fits(int ng) { return gap_of((ng >= 2 ? ng / 2 : 1) * NS - 1) <= D0 && gap_of(ng * NS - 1) <= D1; }
    static constexpr int NG = fits(4) ? 4 : (fits(2) ? 2 : (fits(1) ? 1 : 0));   // 8 ops per gap (NG = 2) measures 4 % slower
    static constexpr bool STAGED = NG > 0;
    static constexpr int GS = STAGED ? 16 / NG : 16;
    static constexpr int slot_at(int gi) {                  // stage slot executed in gap gi, or -1
        if (!usable(gi)) return -1;
        int c = 0;
        for (int g = 0; g < gi; ++g) c += usable(g) ? 1 : 0;
        return c < NG * NS ? c : -1;
    }
};

__device__ __forceinline__ h8 pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 t = {a, b, c, d};
    return __builtin_bit_cast(h8, t);
}

// Stage ST of group G of the plan.
// DOT: the tile's activations also feed a one-row linear layer (_alpha_linear on the last trunk layer's output): dot +=
// w[row] * v for every element, in fp32 on the vector ALU.  The weights of a group (dotw: this tile's row of the table in
// LDS, laid out like a bias row) are read in the group's fma stage and used one per later stage, so that the chain of
// dependent FMAs on `dot` never has two links in one MFMA gap.
template <class P, bool X3, int G, int ST, bool DOT>
__device__ __forceinline__ void epi_stage(const Pend& t, Epi& E, float inv_scale, float lower, h8& y0h, h8& y0l, h8& y1h, h8& y1l,
                                          const float* dotw, float& dot) {
    constexpr int st = ST, e0 = G * P::GS;
    constexpr int ST_PACK = 3, ST_RES = 4, ST_PACKLO = 5, ST_PARK = X3 ? 6 : 4;
    if constexpr (DOT) {
        static_assert(P::NS - ST_PACK >= 1, "no stage left for the dot product");
        constexpr int NDS = P::NS - ST_PACK;                       // stages that carry dot FMAs: ST_PACK .. NS-1
        if (st == 1) {
#pragma unroll
            for (int q = e0 / 4; q < (e0 + P::GS) / 4; ++q) {      // elements 4q..4q+3 = rows 8q + 4h + 0..3 (like Pend::bias)
                const float4 w4 = *reinterpret_cast<const float4*>(dotw + 8 * q);
                E.dw[4 * q] = w4.x; E.dw[4 * q + 1] = w4.y; E.dw[4 * q + 2] = w4.z; E.dw[4 * q + 3] = w4.w;
            }
        }
        if (st >= ST_PACK) {
#pragma unroll
            for (int e = e0; e < e0 + P::GS; ++e)
                if ((e - e0) % NDS == st - ST_PACK) {
                    dot = __builtin_fmaf(E.dw[e], E.v[e], dot);
                    asm volatile("" : "+v"(dot));   // HERE: the sum is only read at the end of the evaluation, and without a use LLVM
                }                                   // sinks every FMA (and keeps every activation alive) down to it
        }
    }
    if (st == 0) {
#pragma unroll
        for (int e = e0; e < e0 + P::GS; ++e) {
            float a = t.a[e];
            asm volatile("" : "+v"(a));   // the accumulator-file read happens HERE, not fused in front of its fma
            E.v[e] = a;
        }
    } else if (st == 1) {
        // (v_pk_fma_f32 on element pairs - half the instructions, the same fma per element - measures 1.4 % SLOWER: 368.3 vs
        // 363.2 ms, alternating on one box; hipcc also needs asm for it and then for the ReLU, whose operand it no longer knows
        // to be canonical)
#pragma unroll
        for (int e = e0; e < e0 + P::GS; ++e) {
            const float4 b = t.bias[e >> 2];
            const float bb = (e & 3) == 0 ? b.x : ((e & 3) == 1 ? b.y : ((e & 3) == 2 ? b.z : b.w));
            E.v[e] = __builtin_fmaf(E.v[e], inv_scale, bb);
        }
    } else if (st == 2) {
#pragma unroll
        for (int e = e0; e < e0 + P::GS; ++e) E.v[e] = fmaxf(E.v[e], lower);
    } else if (st == ST_PACK) {
#pragma unroll
        for (int p = e0 / 2; p < (e0 + P::GS) / 2; ++p) {
            h2 hp;
            hp[0] = (_Float16)E.v[2 * p]; hp[1] = (_Float16)E.v[2 * p + 1];   // one v_cvt_pk_f16_f32
            E.hp[p] = __builtin_bit_cast(uint32_t, hp);
            if (!X3) E.lp[p] = 0u;
        }
    } else if (X3 && st == ST_RES) {
        // residual v - hi as fma(hi, -1, v) with the fp16 half read in place: v_fma_mix_f32 instead of v_cvt_f32_f16 +
        // v_sub_f32 (same single rounding).  hipcc does not select it from C (it folds the -1 into a subtraction first).
#pragma unroll
        for (int p = e0 / 2; p < (e0 + P::GS) / 2; ++p) {
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(E.r[2 * p]) : "v"(E.hp[p]), "v"(E.v[2 * p]));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(E.r[2 * p + 1]) : "v"(E.hp[p]), "v"(E.v[2 * p + 1]));
        }
    } else if (X3 && st == ST_PACKLO) {
#pragma unroll
        for (int p = e0 / 2; p < (e0 + P::GS) / 2; ++p) {
            h2 lp;
            lp[0] = (_Float16)E.r[2 * p]; lp[1] = (_Float16)E.r[2 * p + 1];
            E.lp[p] = __builtin_bit_cast(uint32_t, lp);
        }
    } else if (st == ST_PARK || st == ST_PARK + 1) {
        // Park finished output k-steps in the accumulator half of the register file, where the MFMAs read them directly
        // (as plain VGPR values the allocator moves half of them there anyway and copies each back in front of its use):
        // hi in this gap, lo in the next.  Output k-step 0 is complete with element 7, k-step 1 with element 15.
        const bool lo = st != ST_PARK;
        if (lo && !X3) return;
        const int last = e0 + P::GS - 1;
        if (last == 7 || (P::GS == 16)) {
            if (!lo) { y0h = pack4(E.hp[0], E.hp[1], E.hp[2], E.hp[3]); asm volatile("" : "+a"(y0h)); }
            else     { y0l = pack4(E.lp[0], E.lp[1], E.lp[2], E.lp[3]); asm volatile("" : "+a"(y0l)); }
        }
        if (last == 15) {
            if (!lo) { y1h = pack4(E.hp[4], E.hp[5], E.hp[6], E.hp[7]); asm volatile("" : "+a"(y1h)); }
            else     { y1l = pack4(E.lp[4], E.lp[5], E.lp[6], E.lp[7]); asm volatile("" : "+a"(y1l)); }
        }
    }
}

// Gap GI (0-based over the tile's main k-steps) of the plan: the stage of one group, or nothing.
template <class P, bool X3, int GI, bool DOT>
__device__ __forceinline__ void epi_gap(const Pend& t, Epi& E, float inv_scale, float lower, h8& y0h, h8& y0l, h8& y1h, h8& y1l,
                                        const float* dotw, float& dot) {
    constexpr int slot = P::slot_at(GI);
    if constexpr (slot >= 0) epi_stage<P, X3, slot / P::NS, slot % P::NS, DOT>(t, E, inv_scale, lower, y0h, y0l, y1h, y1l, dotw, dot);
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
// waited for by hand in sync() (s_waitcnt vmcnt(0) + barrier).  Wave w streams the w-th quarter of a chunk (n
// consecutive 1-KiB pieces): source = scalar base + lane*16, so a piece costs scalar instructions only.
//
// M0 carries the wave-uniform LDS destination.  With one wave per SIMD every instruction slot counts (a piece with M0
// saved and restored around it is five), so M0 is OWNED by this kernel: it is written once per group of four pieces and
// left there.  That is sound only while hipcc emits no M0 use of its own in this kernel (it has no reason to on gfx950:
// no movrel, no GDS, no sendmsg) - tests/test_abi.py::test_kernel_owns_m0 greps the generated assembly for exactly that,
// and pieces of a group must be issued in order with no other group in between (tile_mma's static schedule does).
// (An "m0" clobber on the asm statements would say nothing to hipcc: M0 is a reserved register, the clobber is ignored with a
// warning.  tools/check_m0.py is the guard: every M0 write in the disassembly must be the first line of one of these statements,
// and no instruction with an implicit M0 operand may appear in the kernel at all.)
//
// Timeline (tile T consumes chunk T from buffer T&1; PD = fragment prefetch distance in k-steps):
//   * ONE barrier per tile, PD k-steps before the tile's end.  Before it every wave waits for its own LDS reads
//     (all reads of chunk T have been issued by then) and its own DMA pieces (chunk T+1, issued >= 7 k-steps
//     earlier).  After it (a) chunk T+1 is visible, so the A fragments of tile T+1's first PD k-steps are read
//     during the last PD k-steps of tile T and the matrix pipe does not drain at the tile boundary, and (b) buffer
//     T&1 is free, so the DMA of chunk T+2 starts at once: its first PD pieces in tile T, the rest early in T+1.
//   * On the long tiles the pre-barrier wait leaves the two fragment reads issued one k-step earlier in flight
//     (lgkmcnt(2): waiting for them too costs an LDS round trip per tile, 5 % of the frame).  Those two reads fetch the
//     (hi, lo) tiles of the chunk's LAST k-step, and these do not live in the chunk buffer the barrier releases but in
//     one of THREE 2-KiB tail slots, slot = chunk mod 3.  Slot (T+2) mod 3 = (T-1) mod 3 is refilled by DMA pieces that
//     are issued behind the barrier of tile T; its previous content, the tail of chunk T-1, was read one k-step before
//     the barrier of tile T-1 and consumed by every wave's last MFMAs of tile T-1 (a wave waits for a fragment before it
//     multiplies with it), i.e. before that wave ARRIVES at the barrier of tile T.  So no LDS location is ever written
//     while a read of its previous content can be outstanding, whatever the timing: the only reads in flight across a
//     barrier target a slot that no DMA piece issued before the NEXT barrier touches.  (LDS returns a wave's reads in
//     order and nothing else in the tile loop counts on lgkmcnt, so "all but two" is exactly "all but those two".)
template <int CHUNK_BYTES, bool X3>
struct Walker {
    const uint8_t* stream;
    uint32_t next_tile;      // first tile of the next chunk to stream
    uint32_t lds_chunks;     // LDS byte address of chunk buffer 0
    uint32_t lds_tail;       // LDS byte address of tail slot 0 (three slots of two tiles)
    const char* buf0;
    const char* tail0;
    int t3;                  // chunk % 3: tail slot of the chunk being consumed
    uint32_t blk_dst_tail;   // blk_dst for the pieces that go to the tail slot (biased so that piece i lands at base + i KiB)
    int tail_first;          // first piece of this wave's quarter that goes to the tail slot (wave 3 of a long chunk), else huge
    const float* bias_tab;   // LDS bias table of the current network, 32 floats per chunk
    int chunk;               // index of the chunk being consumed
    int b;                   // buffer holding the chunk being consumed
    int wave;
    uint32_t lane_off;       // lane * 16
    const uint8_t* blk_src;  // this wave's quarter of the chunk being streamed (uniform)
    uint32_t blk_dst;
    bool skip_lo = false;    // single-pass mode: the odd pieces of the chunk being streamed are lo tiles
#ifdef NWE_STAMPS
    unsigned long long st_pre = 0, st_wait = 0, st_post = 0, st_t0 = 0;
#endif

    __device__ __forceinline__ void start(const uint8_t* s, const float* bias) {
        stream = s; bias_tab = bias; next_tile = 0; chunk = 0; b = 0; t3 = 0;
    }
    __device__ __forceinline__ const char* cur() const { return buf0 + b * CHUNK_BYTES; }
    __device__ __forceinline__ const char* next() const { return buf0 + (b ^ 1) * CHUNK_BYTES; }
    __device__ __forceinline__ const char* tail() const { return tail0 + t3 * (2 * kTileBytes); }
    // Start streaming a chunk of n_per_wave pieces per wave into `buffer`; ahead = how many chunks it is ahead of the one
    // being consumed (its tail slot is (t3 + ahead) mod 3).
    __device__ __forceinline__ void begin(int n_per_wave, int buffer, int ahead) {
        blk_src = stream + ((size_t)next_tile + (size_t)wave * n_per_wave) * kTileBytes;
        blk_dst = lds_chunks + buffer * CHUNK_BYTES + wave * n_per_wave * kTileBytes;
        next_tile += n_per_wave * kWaves;
        skip_lo = !X3 && (n_per_wave & 1) == 0;
        if (X3) {
            int slot = t3 + ahead;
            slot = slot >= 3 ? slot - 3 : slot;
            const bool lng = n_per_wave >= 8;                               // Shape::LONG_PIECES
            tail_first = (lng && wave == kWaves - 1) ? n_per_wave - 2 : (1 << 20);
            blk_dst_tail = lds_tail + slot * (2 * kTileBytes) - (n_per_wave - 2) * kTileBytes;
        }
    }
    // Piece i of the chunk being streamed.  Pieces go in groups of four: one scalar base per group, the 1-KiB step inside
    // a group rides on the instruction offset, which advances the global source AND the LDS destination (nwe_selftest
    // report[6]).  i is a compile-time constant at every call site.
    // tail_piece: the piece with which this wave's tail pieces would start if the chunk has exactly the caller's static
    // piece count (NB - 2); it rewrites M0 (a no-op for the waves and chunks whose destination does not change there).
    __device__ __forceinline__ void piece(int i, int tail_piece = -1) {
#ifdef NWE_EXP_NODMA   // timing experiments only; a run-time test here would split every k-step into its own basic block
        return;
#endif
        // single-pass mode multiplies by the hi tiles only: where a wave's quarter of the chunk starts on an even tile (all
        // chunks but the view layer's, 9 tiles per wave) the lo tile of every (hi, lo) pair = the odd pieces is neither
        // streamed nor read; the LDS layout keeps its holes.  (A run-time test, but only in the single-pass instantiation.)
        if (!X3 && skip_lo && (i & 1)) return;
        const uint8_t* src = blk_src + (size_t)(i >> 2) * (4 * kTileBytes);
        // the last two pieces of a long chunk (wave 3: n-2, n-1) go to the chunk's tail slot: a scalar select, no branch
        // (a long chunk has >= 8 pieces per wave, so only pieces 6.. can be tail pieces: no select in front of the others)
        const uint32_t base = (X3 && i >= 6 && i >= tail_first) ? blk_dst_tail : blk_dst;
        const uint32_t dst = base + (i >> 2) * (4 * kTileBytes);
        // M0 is written by the first piece of a group and by the piece the tail would start with (6 of an 8-piece quarter, 7
        // of a 9-piece one; the first tail piece of a 10-piece quarter, 8, opens a group anyway, and so does 8 of 9).
        const bool set_m0 = (i & 3) == 0 || (X3 && i == tail_piece);
// Cache policy of the weight stream (NWE_GLDS_POLICY, timing experiments): 0 default, 1 sc1 (bypass the CU's vector L1, which
// never sees a piece twice), 2 nt, 3 sc0 sc1, 4 sc1 nt.
#ifndef NWE_GLDS_POLICY
#define NWE_GLDS_POLICY 0
#endif
#if NWE_GLDS_POLICY == 1
#define NWE_GLDS_POL " sc1"
#elif NWE_GLDS_POLICY == 2
#define NWE_GLDS_POL " nt"
#elif NWE_GLDS_POLICY == 3
#define NWE_GLDS_POL " sc0 sc1"
#elif NWE_GLDS_POLICY == 4
#define NWE_GLDS_POL " sc1 nt"
#else
#define NWE_GLDS_POL ""
#endif
#define NWE_GLDS(OFF) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF NWE_GLDS_POL :: "v"(lane_off), "s"(src) : "memory")
#define NWE_GLDS_M0(OFF) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #OFF NWE_GLDS_POL \
                                      :: "v"(lane_off), "s"(src), "s"(dst) : "memory")
        if (set_m0) {   // point M0 at the group's LDS destination (one wait state before the DMA)
            switch (i & 3) {
                case 0: NWE_GLDS_M0(0); break;
                case 1: NWE_GLDS_M0(1024); break;
                case 2: NWE_GLDS_M0(2048); break;
                default: NWE_GLDS_M0(3072); break;
            }
        } else {
            switch (i & 3) {
                case 1: NWE_GLDS(1024); break;
                case 2: NWE_GLDS(2048); break;
                default: NWE_GLDS(3072); break;
            }
        }
#undef NWE_GLDS
#undef NWE_GLDS_M0
    }
    template <bool LONG_TILE>
    __device__ __forceinline__ void sync() {
#ifdef NWE_EXP_NOSYNC
        return;
#endif
        // Own DMA pieces of chunk T+1 landed and own LDS reads done - on the long tiles EXCEPT the two reads just issued (the
        // fragments of this chunk's last k-step, one k-step ago): waiting for those too costs an LDS round trip per tile
        // (5 % of the frame time).  They read the chunk's tail slot, which this barrier does NOT release (see the
        // timeline above); everything in the chunk buffer it does release has been read.  Short tiles have no tail slot
        // and keep the full wait.
        if (LONG_TILE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    __device__ __forceinline__ void tile_done() { b ^= 1; ++chunk; t3 = t3 == 2 ? 0 : t3 + 1; }
};

// Compile-time loop: f(integral_constant<int, I>) for I in [I0, N).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int PD = 3;   // A fragments are read PD k-steps ahead of their MFMAs
struct Frags { h8 hi[PD + 1], lo[PD + 1]; };   // ring, slot = (k-step counter) mod (PD+1)

// Which k-steps of a tile issue DMA pieces (static; tile_mma and the epilogue plan both read it).  The pieces [PD, NB) of
// chunk T+1 go from k-step 0 on, one per k-step on the long tiles - as early as possible, the barrier at k-step QSYNC =
// NQ - PD waits for them - followed by the two extra pieces of a skip-layer chunk (their slots are reserved whether or not
// the chunk has them); after the barrier come the first min(PD, NA) pieces of chunk T+2.
template <int NB, int NA, int NQ>
struct DmaPlan {
    static constexpr int QSYNC = NQ - PD;
    static constexpr int REST = NB > PD ? NB - PD : 0;
    static constexpr int TOT = REST > 0 ? REST + 2 : 0;                                   // logical slots: pieces, then the two extras
    static constexpr int PPK = TOT == 0 ? 0 : (TOT + (QSYNC > 0 ? QSYNC : 1) - 1) / (QSYNC > 0 ? QSYNC : 1);   // slots per k-step (1 on long tiles)
    static_assert(TOT == 0 || QSYNC > 0, "no k-step in front of the barrier for the DMA pieces");
    // A piece holds the wave's issue for ~30 cycles whatever else the gap carries, and two pieces three MFMAs apart cost more
    // than twice one piece six MFMAs apart (tools/ubench/dma_cost.hip: 20 vs 8.6 cycles each beside 32x32x16 MFMAs), so where
    // the tile is long enough the pieces go out every SECOND k-step: the last one still MARGIN k-steps (~500 cycles, more than
    // an L2-hit LDS-DMA takes to land) in front of the barrier that waits for it; the two extra slots follow back to back.
#ifndef NWE_DMA_STRIDE
#define NWE_DMA_STRIDE 2
#endif
    static constexpr int MARGIN = 5;
    static constexpr int STRIDE = (NWE_DMA_STRIDE == 2 && PPK == 1 && 2 * (REST - 1) <= QSYNC - MARGIN && 2 * (REST - 1) + 2 <= QSYNC - 1) ? 2 : 1;
    static constexpr int kstep_of(int j) { return STRIDE == 1 ? j / (PPK > 0 ? PPK : 1) : (j < REST ? 2 * j : 2 * (REST - 1) + 1 + (j - REST)); }
    static constexpr int lo(int q) {                                                      // slots issued before k-step q
        int n = 0;
        for (int j = 0; j < TOT; ++j) n += kstep_of(j) < q ? 1 : 0;
        return n;
    }
    static_assert(TOT == 0 || lo(QSYNC) == TOT, "every slot must be issued in front of the barrier");
    static constexpr uint32_t mask() {
        uint32_t m = 0;
        for (int q = 0; q < NQ; ++q) {
            const bool pre = q < QSYNC && lo(q + 1) > lo(q);
            const bool post = q >= QSYNC && NA > 0 && q - QSYNC < (NA < PD ? NA : PD);
            if (pre || post) m |= 1u << q;
        }
        return m;
    }
};

// One 32-row tile = NKP optional "pre" k-steps (gamma(x) of the skip layer, taken if use_g) + NKH main k-steps over
// X + NKD "post" k-steps (gamma(d) of the view layer).  Chunk layout in that order, (hi, lo) tile pair per k-step,
// lane-linear.  On entry the fragment ring holds this tile's first PD k-steps in slots PHASE..PHASE+PD-1; on exit
// it holds the next tile's.  The epilogue of the PREVIOUS tile (`prev` -> y*) runs in the MFMA gaps of the main
// k-steps by EpiPlan; FEEDS says that its outputs y* are the last two k-steps of X itself (first tile of a layer, rgb
// head), which sets the plan's deadline.  DMA (DmaPlan): this tile issues the pieces [PD, NB) (+2 if extraB) of chunk
// T+1 in its first k-steps and, after its barrier, pieces [0, min(PD, NA)) of chunk T+2 (NA pieces per wave, +2 if
// extraA; NA = 0: none).  HASNEXT: a tile follows in this pass (its first fragments are prefetched).
// DOT: the pending tile's epilogue also accumulates the dot product of its activations with the row `dot_row` of the dot table
// (32 floats in LDS, see epi_stage) into *dot.
template <int NKP, int NKH, int NKD, int PHASE, bool X3, bool PEND, int NB, int NA, bool HASNEXT, bool FEEDS = false, bool DOT = false, class WalkerT>
__device__ __forceinline__ void tile_mma(WalkerT& wk, Frags& F, int lane, bool use_g, bool extraB, bool extraA, const h8* Ghi,
                                         const h8* Glo, const h8* Xhi, const h8* Xlo, const h8* Dhi, const h8* Dlo, Pend& cur,
                                         const Pend& prev, float inv_scale, float lower, h8& y0h, h8& y0l, h8& y1h, h8& y1l, int na_override = -1,
                                         const float* dot_row = nullptr, float* dot = nullptr) {
    static_assert(!DOT || PEND, "a dot product rides on a pending tile's epilogue");
    float dot_dummy = 0.f;
    float& dot_ref = DOT ? *dot : dot_dummy;
    const float* dotw = DOT ? dot_row + 4 * (lane >> 5) : nullptr;   // this lane half's rows 8q + 4h + i of the 32-float row
    constexpr int R = PD + 1;
    constexpr int NQ = NKH + NKD;            // k-steps after the optional pre segment
    constexpr int QSYNC = NQ - PD;           // the barrier sits in front of this k-step
    static_assert(QSYNC >= 0, "tile too short for the prefetch distance");
    static_assert(NKP % R == 0, "the optional segment must not shift the fragment ring");
    const float4* bp = reinterpret_cast<const float4*>(wk.bias_tab + wk.chunk * 32);
    const int h = lane >> 5;
#ifndef NWE_SPREAD_BIAS
#define NWE_SPREAD_BIAS 1   // the tile's four bias reads ride in the third gaps of k-steps 1..4 instead of all at its start: -0.25 % (375.5 vs 376.5 ms, alternating)
#endif
    constexpr bool SPREAD_BIAS = X3 && NWE_SPREAD_BIAS != 0 && (NKH + NKD) >= 6;
    if constexpr (!SPREAD_BIAS) {
#pragma unroll
        for (int g = 0; g < 4; ++g) cur.bias[g] = bp[2 * g + h];
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    }
    constexpr bool LONG_TILE = X3 && NQ >= 16;   // the (hi, lo) tiles of the last k-step live in the chunk's tail slot (Walker)
    constexpr int S_LONG_PIECES = 8;             // Shape::LONG_PIECES: a chunk of >= 8 pieces per wave is a long one
    const char* cbase = wk.cur() + lane * 16;
    const char* nbase = wk.next() + lane * 16;
    const char* tbase = wk.tail() + lane * 16;
    bool pre_done = false;
    float ekeep = 0.f;   // even element of the epilogue pair in flight (unstaged fallback)
    Epi E;
    if (NKP > 0) {
        if (use_g) {   // pre segment: positions 0..NKP-1 of the chunk; reads stay inside this chunk
#pragma unroll
            for (int s = 0; s < NKP; ++s) {
                const int slot = (PHASE + s + PD) % R;
                F.hi[slot] = *reinterpret_cast<const h8*>(cbase + (2 * (s + PD)) * kTileBytes);
                if (X3) F.lo[slot] = *reinterpret_cast<const h8*>(cbase + (2 * (s + PD) + 1) * kTileBytes);
                const int use = (PHASE + s) % R;
                if (s == 0) {
                    f16v zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.hi[use], Ghi[0], zero, 0, 0, 0);
                    if (X3) {
                        cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.lo[use], Ghi[0], cur.a, 0, 0, 0);
                        cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.hi[use], Glo[0], cur.a, 0, 0, 0);
                    }
                } else {
                    mma3<X3>(F.hi[use], F.lo[use], Ghi[s], Glo[s], cur.a);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, X3 ? 2 : 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, X3 ? 3 : 1, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            cbase += NKP * 2 * kTileBytes;
            pre_done = true;
        }
    }
    static_for<0, NQ>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        const h8* Xh = q < NKH ? &Xhi[q] : &Dhi[q - NKH];
        const h8* Xl = q < NKH ? &Xlo[q] : &Dlo[q - NKH];
        if (q == QSYNC) {
#ifdef NWE_STAMPS
            { const unsigned long long t = __builtin_amdgcn_s_memtime(); wk.st_pre += t - wk.st_t0; wk.st_t0 = t; }
#endif
            wk.template sync<LONG_TILE>();
#ifdef NWE_STAMPS
            { const unsigned long long t = __builtin_amdgcn_s_memtime(); wk.st_wait += t - wk.st_t0; wk.st_t0 = t; }
#endif
            if (NA > 0) wk.begin(na_override >= 0 ? na_override : NA + (extraA ? 2 : 0), wk.b, 2);
        }
        // first MFMA of the k-step (hi.hi); everything else of the k-step is issued behind it, while it executes
        const int use = (PHASE + q) % R;
#ifndef NWE_ONE_WAIT
#define NWE_ONE_WAIT 1   // both fragments of the k-step are "used" in front of its first MFMA, so hipcc waits for them once (lgkmcnt)
#endif                   // instead of once per MFMA that consumes one: 16 fewer s_waitcnt per long tile, -0.35 % (372.2 vs 373.5 ms); 0 = off
        if constexpr (X3 && NWE_ONE_WAIT != 0) asm volatile("" :: "v"(F.hi[use]), "v"(F.lo[use]));
        if (q == 0 && !(NKP > 0 && pre_done)) {
            f16v zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.hi[use], *Xh, zero, 0, 0, 0);
        } else {
            cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.hi[use], *Xh, cur.a, 0, 0, 0);
        }
        // DMA piece of this k-step (DmaPlan).  In the three-pass kernel it goes behind the SECOND MFMA: a piece costs ~16
        // issue cycles and next to the two fragment reads it would overrun the 32 cycles of the MFMA it hides behind.
        auto dma = [&]() __attribute__((always_inline)) {
            using DP = DmaPlan<NB, NA, NQ>;
            if constexpr (q < QSYNC && DP::TOT > 0) {
#pragma unroll
                for (int j = DP::lo(q); j < DP::lo(q + 1); ++j) {
                    if (j < DP::REST) wk.piece(PD + j, NB >= S_LONG_PIECES ? NB - 2 : -1);
                    else if (extraB) wk.piece(NB + j - DP::REST);
                }
            }
            if constexpr (q >= QSYNC && NA > 0 && q - QSYNC < (NA < PD ? NA : PD)) wk.piece(q - QSYNC);
        };
        if constexpr (!X3) dma();
        // fragment read of position q+PD: this chunk, or the next tile's first k-steps (visible since the barrier)
#ifndef NWE_SPLIT_READS
#define NWE_SPLIT_READS 1   // the lo fragment is read in the k-step's THIRD gap, not beside the hi fragment: -0.45 % (362.4 vs 364.1 ms, alternating); 0 = both behind the first MFMA
#endif
        auto read_frag = [&](bool want_hi, bool want_lo) __attribute__((always_inline)) {
            if (LONG_TILE && q + PD == NQ - 1) {
                const int slot = (PHASE + q + PD) % R;   // the two reads that stay in flight across the barrier: the tail slot
                if (want_lo) F.lo[slot] = *reinterpret_cast<const h8*>(tbase + kTileBytes);
                if (want_hi) F.hi[slot] = *reinterpret_cast<const h8*>(tbase);
            } else if (q + PD < NQ) {
                const int slot = (PHASE + q + PD) % R;   // lo first: the first MFMA of the k-step needs hi, so one wait covers both
                if (X3 && want_lo) F.lo[slot] = *reinterpret_cast<const h8*>(cbase + (2 * (q + PD) + 1) * kTileBytes);
                if (want_hi) F.hi[slot] = *reinterpret_cast<const h8*>(cbase + (2 * (q + PD)) * kTileBytes);
            } else if (HASNEXT) {
                const int slot = (PHASE + q + PD) % R;
                if (X3 && want_lo) F.lo[slot] = *reinterpret_cast<const h8*>(nbase + (2 * (q + PD - NQ) + 1) * kTileBytes);
                if (want_hi) F.hi[slot] = *reinterpret_cast<const h8*>(nbase + (2 * (q + PD - NQ)) * kTileBytes);
            }
        };
        constexpr bool SPLIT_RD = X3 && NWE_SPLIT_READS != 0;
        read_frag(true, !SPLIT_RD);
        using Plan = EpiPlan<X3, NKH, NQ, FEEDS, DmaPlan<NB, NA, NQ>::mask()>;
        constexpr int GPK = Plan::GPK;
#ifdef NWE_EXP_NOEPI   // timing experiment: no epilogue at all (results are garbage); the pending accumulator is kept alive
        if constexpr (PEND && q == 0) asm volatile("" :: "a"(prev.a));
#define NWE_EPI_ON false
#else
#define NWE_EPI_ON true
#endif
        if constexpr (NWE_EPI_ON && PEND && Plan::STAGED) epi_gap<Plan, X3, GPK * q, DOT>(prev, E, inv_scale, lower, y0h, y0l, y1h, y1l, dotw, dot_ref);
        if (NWE_EPI_ON && PEND && !Plan::STAGED && q < NKH) {
            // short tiles whose outputs feed their own last k-steps have no room for the staged plan: element e runs in
            // k-step floor(e*(NKH-1)/16), so all sixteen are done one k-step before the tile's last
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((e * (NKH - 1)) / 16 == q) {
                    finish_elem<X3, DOT>(prev, e, inv_scale, lower, ekeep, y0h, y0l, y1h, y1l, dotw, &dot_ref);
                    if (e == 7) asm volatile("" : "+a"(y0h), "+a"(y0l));
                }
        }
        // Issue order: each MFMA opens its own scheduling region (hard fence behind every gap), the fragment reads and the
        // DMA piece follow the first one.  Inside a region the ops are independent of each other by construction, so the
        // order hipcc picks there costs nothing; without the fences it sinks the prefetch reads (issued PD k-steps early
        // on purpose) to their first use and clusters the epilogue into dependent chains at the end of the tile.
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q + PD < NQ || HASNEXT) __builtin_amdgcn_sched_group_barrier(0x100, (X3 && !SPLIT_RD) ? 2 : 1, 0);
        if (X3) {
            __builtin_amdgcn_sched_barrier(0);
            cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.lo[use], *Xh, cur.a, 0, 0, 0);
            dma();
            if constexpr (NWE_EPI_ON && PEND && Plan::STAGED) epi_gap<Plan, X3, GPK * q + 1, DOT>(prev, E, inv_scale, lower, y0h, y0l, y1h, y1l, dotw, dot_ref);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            cur.a = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.hi[use], *Xl, cur.a, 0, 0, 0);
            if constexpr (SPLIT_RD) read_frag(false, true);
            if constexpr (SPREAD_BIAS && q >= 1 && q <= 4) cur.bias[q - 1] = bp[2 * (q - 1) + h];
            if constexpr (NWE_EPI_ON && PEND && Plan::STAGED) epi_gap<Plan, X3, GPK * q + 2, DOT>(prev, E, inv_scale, lower, y0h, y0l, y1h, y1l, dotw, dot_ref);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (SPLIT_RD && (q + PD < NQ || HASNEXT)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (SPREAD_BIAS && q >= 1 && q <= 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // Keep the epilogue HERE: its results are only consumed by the next layer, so without a use at this point
    // LLVM sinks the whole epilogue of every tile of a layer to the layer's end (and keeps all their accumulators
    // alive), which is exactly the un-overlapped VALU block this structure is meant to remove.
    // The "a" constraint also parks the finished fragments in the accumulator half of the register file, where the
    // MFMAs read them directly; as plain VGPR values the allocator spills half of them there anyway and copies each
    // back (4 v_accvgpr_read + s_nop) in front of every MFMA that uses it.
    if (PEND) asm volatile("" : "+a"(y1h), "+a"(y1l));
#ifdef NWE_STAMPS
    { const unsigned long long t = __builtin_amdgcn_s_memtime(); wk.st_post += t - wk.st_t0; wk.st_t0 = t; }
#endif
    wk.tile_done();
}

// A full layer of NT tiles reading X (+ gamma k-steps) and writing Y.  Tile rt accumulates into P[rt&1] while the
// epilogue of the tile before it runs: for rt = 0 that is the LAST tile of the previous layer (in P1, destined for
// k-steps 2*NT-2, 2*NT-1 of X itself), for rt > 0 tile rt-1 of this layer (destined for Y).  On return P1 holds
// this layer's last tile, still pending.  Chunk sizes for the DMA schedule, in pieces per wave: this layer's chunks
// N_THIS (+2 when use_g), the following layer's N_AFTER (+2 when extra_after), and `first_nb` = what tile 0 still
// has to issue of chunk T+1 (0 at the very start of a pass, where chunks 0 and 1 are streamed up front).  NA_LAST: what the
// layer's last tile starts of chunk T+2 - N_AFTER unless only ONE chunk follows the layer (kFormNoViewDirs: 0).
template <int NT, int NKP, int NKH, bool X3, bool PEND0, int N_AFTER, bool PASS_START, bool DOT = false, int NA_LAST = N_AFTER, class WalkerT>
__device__ __forceinline__ void layer(WalkerT& wk, Frags& F, int lane, bool use_g, bool extra_after, const h8* Ghi, const h8* Glo,
                                      h8* Xhi, h8* Xlo, h8* Yhi, h8* Ylo, Pend& P0, Pend& P1, float inv_scale, float lower_prev,
                                      float lower, int na_last_override = -1, const float* dot_tab = nullptr, float* dot = nullptr) {
    static_assert(NT % 2 == 0 && NT >= 4, "tiles per layer must be even (accumulator ping-pong)");
    constexpr int N_THIS = 2 * NKH / kWaves;
    // DOT: this layer's activations also feed a one-row linear layer; tile rt's share is accumulated with its epilogue, i.e. in
    // tile rt + 1 (row rt of dot_tab); the last tile's share rides on the epilogue the caller runs in the tile after the layer.
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        Pend& cur = (rt & 1) ? P1 : P0;
        Pend& prev = (rt & 1) ? P0 : P1;
        // chunk T+1 / T+2 seen from tile rt: inside the layer both are this layer's; at its end the next layer's
        const bool ebB = rt + 1 < NT ? use_g : extra_after;
        const bool ebA = rt + 2 < NT ? use_g : extra_after;
        const float* drow = DOT ? dot_tab + (rt - 1) * 32 : nullptr;
        if (rt == 0) {
            constexpr int NB0 = PASS_START ? 0 : N_THIS;
            if constexpr (PEND0) {
                constexpr int L = 2 * NT - 2;   // the previous layer has as many tiles as X has k-step pairs
                tile_mma<NKP, NKH, 0, 0, X3, true, NB0, N_THIS, true, true>(wk, F, lane, use_g, ebB, ebA, Ghi, Glo, Xhi, Xlo, nullptr, nullptr,
                                                                      cur, prev, inv_scale, lower_prev, Xhi[L], Xlo[L], Xhi[L + 1], Xlo[L + 1]);
            } else {
                h8 d0, d1, d2, d3;
                tile_mma<NKP, NKH, 0, 0, X3, false, NB0, N_THIS, true>(wk, F, lane, use_g, ebB, ebA, Ghi, Glo, Xhi, Xlo, nullptr, nullptr,
                                                                       cur, prev, inv_scale, lower_prev, d0, d1, d2, d3);
            }
        } else if (rt + 2 < NT) {
            tile_mma<NKP, NKH, 0, 0, X3, true, N_THIS, N_THIS, true, false, DOT>(wk, F, lane, use_g, ebB, ebA, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur,
                                                                     prev, inv_scale, lower, Yhi[2 * rt - 2], Ylo[2 * rt - 2], Yhi[2 * rt - 1],
                                                                     Ylo[2 * rt - 1], -1, drow, dot);
        } else if (rt + 1 < NT) {
            tile_mma<NKP, NKH, 0, 0, X3, true, N_THIS, N_AFTER, true, false, DOT>(wk, F, lane, use_g, ebB, ebA, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur,
                                                                      prev, inv_scale, lower, Yhi[2 * rt - 2], Ylo[2 * rt - 2], Yhi[2 * rt - 1],
                                                                      Ylo[2 * rt - 1], -1, drow, dot);
        } else {
            tile_mma<NKP, NKH, 0, 0, X3, true, N_AFTER, NA_LAST, true, false, DOT>(wk, F, lane, use_g, ebB, ebA, Ghi, Glo, Xhi, Xlo, nullptr, nullptr, cur,
                                                                       prev, inv_scale, lower, Yhi[2 * rt - 2], Ylo[2 * rt - 2], Yhi[2 * rt - 1],
                                                                       Ylo[2 * rt - 1], na_last_override, drow, dot);
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
    const float first = h ? (float)(1 << NB) : 1.f;   // 2^(NB*h): this lane half's lowest octave
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = c == 0 ? vx : (c == 1 ? vy : vz);
        float sn[NB], cs[NB];
        octave_sincos<NB>(v, first, sn, cs);             // embedding.py:36: fn(x * freq) for freq = first * 2^bl
#pragma unroll
        for (int bl = 0; bl < NB; ++bl) {
            vals[2 * (bl * 3 + c)] = sn[bl];
            vals[2 * (bl * 3 + c) + 1] = cs[bl];
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

// View-layer tiles RT..NTV-1 (compile-time recursion: the DMA schedule and the ring phase depend on RT).  Each tile has
// KH + KD k-steps, which shifts the fragment ring by (KH+KD) mod (PD+1) per tile.
// SIGMA_TILE (the unfolded formulation): tile 0 follows the alpha tile, which has no activation output to finish; tile RT
// accumulates in P[(RT+1)&1].  !SIGMA_TILE (FOLD): tile 0 follows the last trunk tile directly (pending in P1) and runs its
// epilogue - ReLU into the last two k-steps of X itself, and the last share of the alpha dot product (row NT-1 of dot_tab) -
// so tile RT accumulates in P[RT&1].
template <int RT, int W, int D, bool X3, bool SIGMA_TILE, class WalkerT>
__device__ __forceinline__ void view_tiles(WalkerT& wk, Frags& F, int lane, h8* Ahi, h8* Alo, const h8* GDhi,
                                           const h8* GDlo, h8* Bhi, h8* Blo, Pend& P0, Pend& P1, float inv_scale,
                                           const float* dot_tab = nullptr, float* dot = nullptr) {
    using S = Shape<W, D>;
    constexpr int par = SIGMA_TILE ? (RT + 1) & 1 : RT & 1;
    Pend& cur = par ? P1 : P0;
    Pend& prev = par ? P0 : P1;
    constexpr int PH = (RT * (S::KH + S::KD)) % (PD + 1);
    constexpr int NB = RT + 1 < S::NTV ? S::N_V : S::N_RGB;                            // chunk T+1
    constexpr int NA = RT + 2 < S::NTV ? S::N_V : (RT + 2 == S::NTV ? S::N_RGB : 0);   // chunk T+2
    if constexpr (RT == 0 && SIGMA_TILE) {
        h8 d0, d1, d2, d3;
        tile_mma<0, S::KH, S::KD, PH, X3, false, NB, NA, true>(wk, F, lane, false, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo, cur,
                                                               prev, inv_scale, 0.f, d0, d1, d2, d3);
    } else if constexpr (RT == 0) {
        constexpr int L = 2 * S::NT - 2;
        tile_mma<0, S::KH, S::KD, PH, X3, true, NB, NA, true, true, true>(wk, F, lane, false, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo,
                                                                          cur, prev, inv_scale, 0.f, Ahi[L], Alo[L], Ahi[L + 1], Alo[L + 1], -1,
                                                                          dot_tab + (S::NT - 1) * 32, dot);
    } else {
        tile_mma<0, S::KH, S::KD, PH, X3, true, NB, NA, true>(wk, F, lane, false, false, false, nullptr, nullptr, Ahi, Alo, GDhi, GDlo, cur,
                                                              prev, inv_scale, 0.f, Bhi[2 * RT - 2], Blo[2 * RT - 2], Bhi[2 * RT - 1],
                                                              Blo[2 * RT - 1]);
    }
    if constexpr (RT + 1 < S::NTV) view_tiles<RT + 1, W, D, X3, SIGMA_TILE>(wk, F, lane, Ahi, Alo, GDhi, GDlo, Bhi, Blo, P0, P1, inv_scale);
}

// One MLP evaluation for the wave's 32 points.  nerf/models/nerf_model.py:45-83.
// Trunk layers run as pairs A->B, B->A so that the two activation register sets keep fixed names inside a rolled loop;
// every tile's epilogue is deferred into the next tile (see Pend).
//
// FOLD (the product path): _feature_linear has no activation (nerf_model.py:64) and feeds only the view layer (:66-70), so
// the packer multiplies it into the view layer's weights (nwe_abi.hip: pack_mfma): trunk layers 1..D-1 = D/2 - 1 pairs and
// one single layer A->B, then _alpha_linear and the folded view layer both read B = h, the rgb head reads the view layer's
// output in A.  !FOLD evaluates the feature layer as the reference formulates it (D/2 pairs, the last pair's second layer
// is the feature layer without ReLU; alpha reads B, the view layer A); kept selectable for comparison.
// kFormNoViewDirs (use_view_dirs=False, nerf_model.py:42-43,78-79): the trunk as in FOLD (D/2 - 1 pairs and the single last
// layer A -> B, without the dot product), then ONE tile of _output_linear on B = h whose rows 0..3 are rgb_raw, sigma_raw
// (copies in rows 4..7 for the upper lane half; the reference ignores the fifth channel too: model_utils.py:62,71).
// On entry chunks 0 and 1 of the stream are visible / in flight and F holds the first PD k-steps of chunk 0.
template <int W, int D, int SKIP, bool X3, int FORM, class WalkerT>
__device__ __forceinline__ void mlp_eval(WalkerT& wk, Frags& F, int lane, float inv_scale, h8* Ghi, h8* Glo, const char* gd_lds,
                                         const float* dot_tab, float& o_r, float& o_g, float& o_b, float& o_s) {
    using S = Shape<W, D>;
    static_assert(D % 2 == 0, "trunk depth must be even");
    static_assert(SKIP < 0 || SKIP % 2 == 0, "skip layer index must be even");
    static_assert(S::NT % 2 == 0 && S::NTV % 2 == 0, "tile counts must be even");
    constexpr bool FOLD = FORM == kFormFolded, NOVIEW = FORM == kFormNoViewDirs;
    h8 Ahi[S::KH], Alo[S::KH], Bhi[S::KH], Blo[S::KH];
    Pend P0, P1;
    constexpr int NPAIR = D / 2;
    constexpr int SKIP_PAIR = SKIP < 0 ? -1 : SKIP / 2;   // pair whose first layer takes [gamma, h]

    // layer 0: gamma(x) -> A (nothing pending in front of its first tile); the layer after it opens pair 0
    layer<S::NT, 0, S::KG, X3, false, S::N_H, true>(wk, F, lane, false, SKIP_PAIR == 0, nullptr, nullptr, Ghi, Glo, Ahi, Alo, P0, P1,
                                                    inv_scale, 0.f, 0.f);
    // One pair of trunk layers.  use_g / skip_next are literals at every call site, so the run-time tests on them inside
    // tile_mma (the optional gamma(x) k-steps, the two extra DMA pieces of a skip-layer chunk) fold away per site.
    auto pair_body = [&](bool use_g, bool skip_next, bool last) __attribute__((always_inline)) {
        // first of pair: (gamma +) A -> B, ReLU.  Its first tile finishes the pending last tile of A (ReLU: the
        // producer is layer 0 or a non-final second-of-pair layer).
        layer<S::NT, S::KG, S::KH, X3, true, S::N_H, false>(wk, F, lane, use_g, false, Ghi, Glo, Ahi, Alo, Bhi, Blo, P0, P1, inv_scale,
                                                            0.f, 0.f);
        // second of pair: B -> A; !FOLD: the last pair's second layer is _feature_linear (no ReLU, nerf_model.py:64).
        // After it comes the next pair's first layer (skip: 2 more pieces) or the alpha tile and then the view layer.
        layer<S::NT, 0, S::KH, X3, true, S::N_H, false>(wk, F, lane, false, skip_next, nullptr, nullptr, Bhi, Blo,
                                                        Ahi, Alo, P0, P1, inv_scale, 0.f, (FORM == kFormReference && last) ? -INFINITY : 0.f,
                                                        (FORM == kFormReference && last) ? S::N_V : -1);
    };
    constexpr int PAIRS = FORM == kFormReference ? NPAIR : NPAIR - 1;   // pairs evaluated here (otherwise the last trunk layer stands alone below)
#ifndef NWE_PEEL_SKIP
#define NWE_PEEL_SKIP 1   // the pair that takes gamma(x) is peeled out of the rolled loop: no run-time use_g tests inside the tiles (6 branches per tile of every first-of-pair layer), +16 tiles of code (110 KB): -0.8 % (364.4 vs 367.2 ms, alternating); 0 = one rolled loop
#endif
    if constexpr (NWE_PEEL_SKIP != 0 && SKIP_PAIR >= 0 && SKIP_PAIR < PAIRS) {
#pragma unroll 1
        for (int pair = 0; pair < SKIP_PAIR; ++pair) pair_body(false, pair + 1 == SKIP_PAIR, false);
        pair_body(true, false, SKIP_PAIR == NPAIR - 1);
#pragma unroll 1
        for (int pair = SKIP_PAIR + 1; pair < PAIRS; ++pair) pair_body(false, false, pair == NPAIR - 1);
    } else {
#pragma unroll 1
        for (int pair = 0; pair < PAIRS; ++pair) {
            const bool last = pair == NPAIR - 1;
            pair_body(pair == SKIP_PAIR, !last && pair + 1 == SKIP_PAIR, last);
        }
    }
    float sig = 0.f;   // FOLD: this lane half's share of _alpha_linear . h
    if constexpr (FOLD) {
        // FOLD: the last trunk layer stands alone (A -> B); behind it comes the view layer at once (its chunks are N_V pieces).
        // _alpha_linear (nerf_model.py:63) is one output row on this layer's activations h: sigma = w . h + b is accumulated
        // in fp32 on the vector ALU with the tiles' epilogues (row rt of dot_tab holds w[32 rt .. 32 rt + 31]) instead of a
        // 32-row MFMA tile of which one row would be used (48 of 3168 MFMAs, 16 KB of the weight stream per evaluation).
        constexpr bool G_LAST = SKIP_PAIR == NPAIR - 1;
        layer<S::NT, G_LAST ? S::KG : 0, S::KH, X3, true, S::N_V, false, true>(wk, F, lane, G_LAST, false, Ghi, Glo, Ahi, Alo, Bhi, Blo, P0, P1,
                                                                               inv_scale, 0.f, 0.f, -1, dot_tab, &sig);
    }
    constexpr int L = 2 * S::NT - 2;
    constexpr int LV = 2 * S::NTV - 2;
    if constexpr (NOVIEW) {
        // the last trunk layer A -> B: ONE chunk follows it (N_H pieces per wave), so its last tile starts no chunk T+2
        constexpr bool G_LAST = SKIP_PAIR == NPAIR - 1;
        layer<S::NT, G_LAST ? S::KG : 0, S::KH, X3, true, S::N_H, false, false, 0>(wk, F, lane, G_LAST, false, Ghi, Glo, Ahi, Alo, Bhi, Blo, P0, P1,
                                                                                   inv_scale, 0.f, 0.f);
        // _output_linear in P0 while the last trunk tile (P1, NT even) is finished - ReLU - into the last two k-steps of B, which
        // this tile itself reads (FEEDS).  Nothing is streamed behind it: the caller starts the next pass.
        tile_mma<0, S::KH, 0, 0, X3, true, 0, 0, false, true>(wk, F, lane, false, false, false, nullptr, nullptr, Bhi, Blo, nullptr, nullptr, P0,
                                                              P1, inv_scale, 0.f, Bhi[L], Blo[L], Bhi[L + 1], Blo[L + 1]);
        o_r = pend_value(P0, 0, inv_scale);
        o_g = pend_value(P0, 1, inv_scale);
        o_b = pend_value(P0, 2, inv_scale);
        o_s = pend_value(P0, 3, inv_scale);
        return;
    }
    static_assert((S::NTV * (S::KH + S::KD)) % (PD + 1) == 0, "the view tiles must restore the ring phase");
    // gamma(d) is per-ray, used by the view layer only: it waits in LDS (this lane's 16 bytes of each fragment tile) instead
    // of holding 16 registers through the trunk.  Read behind the trunk's last barrier, long before the view tiles' k-steps
    // KH.. need it; older than the fragment reads the tile barriers leave in flight.
    h8 GDhi[S::KD], GDlo[S::KD];
#pragma unroll
    for (int k = 0; k < S::KD; ++k) {
        GDhi[k] = *reinterpret_cast<const h8*>(gd_lds + (2 * k) * kTileBytes);
        if (X3) GDlo[k] = *reinterpret_cast<const h8*>(gd_lds + (2 * k + 1) * kTileBytes);
    }
    if constexpr (FOLD) {
        // folded view layer: [h (B), gamma(d)] -> A[0..KV), ReLU (nerf_model.py:64-70 with W_v[:, :W] . W_f multiplied out).
        // Its first tile runs the epilogue of the last trunk tile (P1, NT even): ReLU into the last two k-steps of B itself and
        // the last share of the alpha dot product.  Tile RT accumulates in P[RT & 1], so the last one (NTV even) is in P1.
        view_tiles<0, W, D, X3, false>(wk, F, lane, Bhi, Blo, GDhi, GDlo, Ahi, Alo, P0, P1, inv_scale, dot_tab, &sig);
        // both lane halves hold half of the features: the other half's share comes over the 32-lane swap; the row behind the
        // weights holds the bias in element 0
        const float sigma = __fadd_rn(__fadd_rn(sig, __shfl_xor(sig, 32, 64)), dot_tab[S::NT * 32]);
        // rgb head (nerf_model.py:74) in P0 while the last view tile (P1) is finished into A; rows 0..2 and their copies 4..6
        // for the upper lane half.  Nothing is streamed behind it: the caller starts the next pass.
        tile_mma<0, S::KV, 0, 0, X3, true, 0, 0, false, true>(wk, F, lane, false, false, false, nullptr, nullptr, Ahi, Alo, nullptr, nullptr, P0,
                                                              P1, inv_scale, 0.f, Ahi[LV], Alo[LV], Ahi[LV + 1], Alo[LV + 1]);
        o_s = sigma;
        o_r = pend_value(P0, 0, inv_scale);
        o_g = pend_value(P0, 1, inv_scale);
        o_b = pend_value(P0, 2, inv_scale);
        return;
    } else {
        // _alpha_linear on B, the input of _feature_linear (nerf_model.py:63); meanwhile the last feature tile (P1) is
        // finished into A without ReLU.  Rows 0 and 4 of the alpha tile both hold the single output row.
        tile_mma<0, S::KH, 0, 0, X3, true, S::N_V, S::N_V, true>(wk, F, lane, false, false, false, nullptr, nullptr, Bhi, Blo, nullptr, nullptr,
                                                                 P0, P1, inv_scale, -INFINITY, Ahi[L], Alo[L], Ahi[L + 1], Alo[L + 1]);
        const float sigma = pend_value(P0, 0, inv_scale);
        // view layer: [feature (A), gamma(d)] -> B[0..KV), ReLU (nerf_model.py:66-70)
        view_tiles<0, W, D, X3, true>(wk, F, lane, Ahi, Alo, GDhi, GDlo, Bhi, Blo, P0, P1, inv_scale);
        tile_mma<0, S::KV, 0, 0, X3, true, 0, 0, false, true>(wk, F, lane, false, false, false, nullptr, nullptr, Bhi, Blo, nullptr, nullptr, P1,
                                                              P0, inv_scale, 0.f, Bhi[LV], Blo[LV], Bhi[LV + 1], Blo[LV + 1]);
        o_s = sigma;
    }
    o_r = pend_value(P1, 0, inv_scale);
    o_g = pend_value(P1, 1, inv_scale);
    o_b = pend_value(P1, 2, inv_scale);
}

// Coarse samples the LDS weight / cdf buffer holds: four packets per workgroup keep one buffer per wave (64 samples); the
// sample-split decomposition has ONE packet per workgroup and one shared buffer, which holds the ABI's full 128 samples
// in half the space - so more than 64 coarse samples always take that decomposition (launch_t).
constexpr int kPacketMaxSamples = 64;
constexpr int kSplitMaxSamples = kMaxSamples;

template <int W, int D, bool SPLIT>
struct Smem {
    using S = Shape<W, D>;
    static constexpr int WBYTES = (SPLIT ? kSplitMaxSamples : kWaves * kPacketMaxSamples) * kRaysPerWave * 4;
    static constexpr int CHUNKS = 2 * S::CHUNK_BYTES;
    static constexpr int BOFF = CHUNKS;                                              // bias tables, coarse then fine
    static constexpr int BIAS_BYTES = ((S::N_CHUNKS * 32 * 4 + 255) / 256) * 256;
    static constexpr int WOFF = BOFF + 2 * BIAS_BYTES;                               // per-wave coarse weights / cdf
    static constexpr int TOFF = WOFF + WBYTES;                                       // t, 1-t, u tables
    static constexpr int XOFF = TOFF + (2 * kMaxSamples + kMaxImportance) * 4;         // sample-split mode: shaded samples, 2 buffers
    static constexpr int LOFF = XOFF + (SPLIT ? 2 * kWaves * kRaysPerWave * 16 : 0);   // three tail slots of two tiles (Walker)
    static constexpr int GOFF = LOFF + 3 * 2 * kTileBytes;                           // gamma(d) fragments, (hi, lo) per k-step and wave
    static constexpr int TOTAL = GOFF + kWaves * 2 * S::KD * kTileBytes;
    static_assert(XOFF % 16 == 0 && LOFF % 16 == 0 && GOFF % 16 == 0 && TOTAL <= 160 * 1024, "LDS budget");
};

// Two work decompositions, same arithmetic in the same order (results are bit-identical):
//   SPLIT = false: the four waves of a workgroup own four ray packets (128 rays) and walk all their samples;
//   SPLIT = true:  the workgroup owns ONE packet (32 rays); wave w evaluates samples 4i + w, the shaded samples (colour,
//                  opacity) are exchanged through LDS and every wave runs the sequential compositing / importance
//                  sampling for all samples (a few dozen VALU ops per sample, redundantly).  The scheduling unit is a
//                  quarter of the rays and a quarter of the iterations: a 320x240 frame fills the last round of
//                  workgroups 17 % better, a 64x64 frame runs 3x faster; launch_t() picks per launch.
// What a plain frame does not use: everything but rgb / depth / acc / flags of pinhole views - the coarse-pass and
// diagnostic outputs, raw network outputs, sample depths, coarse weights, the test hooks, the training-mode tables,
// precomputed rays.  A launch without any of them takes the LEAN instantiation, in which they are compile-time null: their
// ~40 pointers otherwise sit in (and spill from) the scalar registers of a kernel that has none to spare - 4.4 GB of
// scratch writes per 800x800 frame before this split (profiles/r02_pmc_summary.txt).
__host__ __device__ inline bool is_lean(const RenderArgs& a) {
    const nwe_outputs& o = a.out;
    return !o.raw_coarse && !o.raw_fine && !o.z_fine && !o.weights_coarse && !o.disp && !o.z_std && !o.rgb_coarse && !o.depth_coarse &&
           !o.acc_coarse && !o.disp_coarse && !o.sample_cond && !o.sample_amp && !o.sample_switch && !o.feat_map && !a.z_fine_in && !a.raw_in_c &&
           !a.raw_in_f && !a.w_in && !a.t_rand && !a.noise_c && !a.noise_f && !a.u_rand && !a.stamps && !a.rays;
}

template <int W, int D, int SKIP, bool X3, bool SPLIT, int FORM, bool LEAN>
__global__ void __launch_bounds__(256) render_mfma_kernel(RenderArgs a_in, NetMfma nc, NetMfma nf) {
    RenderArgs a = a_in;
    if constexpr (LEAN) {
        a.out.raw_coarse = a.out.raw_fine = a.out.z_fine = a.out.weights_coarse = nullptr;
        a.out.disp = a.out.z_std = a.out.rgb_coarse = a.out.depth_coarse = a.out.acc_coarse = a.out.disp_coarse = nullptr;
        a.out.sample_cond = a.out.sample_amp = a.out.sample_switch = a.out.feat_map = nullptr;
        a.z_fine_in = a.raw_in_c = a.raw_in_f = a.w_in = a.t_rand = a.noise_c = a.noise_f = a.u_rand = nullptr;
        a.stamps = nullptr; a.rays = nullptr;
    }
    using S = Shape<W, D>;
    using SM = Smem<W, D, SPLIT>;
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
    float* s_bias = reinterpret_cast<float*>(smem + SM::BOFF);
    constexpr int NCH = S::n_chunks(FORM);       // the launcher checks n_chunks of both networks against it
    constexpr int NROWS = S::n_bias_rows(FORM);  // kFormFolded: the alpha layer's weights and bias ride behind the bias rows
    static_assert(NROWS * 32 * 4 <= SM::BIAS_BYTES, "bias table too small for the dot rows");
    for (int i = threadIdx.x; i < NROWS * 32; i += 256) {
        s_bias[i] = nc.bias[i];
        if (ni > 0) s_bias[SM::BIAS_BYTES / 4 + i] = nf.bias[i];
    }

    const int64_t packet = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * kWaves + wave;
    const int64_t ridx64 = a.ray_first + packet * kRaysPerWave + (lane & 31);
    const bool lane_live = ridx64 < a.n_rays && half == 0;    // this lane stores per-sample outputs of its ray
    const bool live = lane_live && (!SPLIT || wave == 0);      // ... and the per-ray results (every wave holds them in SPLIT mode)
    // One 32-bit row index per lane (the ABI keeps n_rays below 2^31): the ray's own index, or the call's last ray for the
    // lanes of a ragged last packet, which compute along and store nothing.  64-bit only where an offset is formed.
    const int row = (int)(ridx64 < a.n_rays ? ridx64 : a.n_rays - 1);
    const int64_t ridx = row, rclamp = row;
    // The ray is kept as its three-register seed and expanded at the top of every sample iteration (bit-identical by
    // construction): nothing of it but |d| stays in registers across an MLP evaluation.  The empty asm hides the seed from
    // loop-invariant code motion, which would otherwise hoist the expansion and spill its results.
    const RaySeed seed = seed_ray(a, rclamp);
    auto fresh_ray = [&]() __attribute__((always_inline)) {
        RaySeed sd = seed;
        asm volatile("" : "+v"(sd.pose), "+v"(sd.x), "+v"(sd.y));
        return make_ray<false>(a, sd);
    };

    Walker<S::CHUNK_BYTES, X3> wk;
    wk.buf0 = smem; wk.lds_chunks = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    wk.tail0 = smem + SM::LOFF; wk.lds_tail = wk.lds_chunks + SM::LOFF; wk.t3 = 0;
    wk.b = 0; wk.wave = wave; wk.lane_off = lane * 16;

    // gamma(d): once per ray (model_utils.py:23-25 re-embeds the same direction for every sample), parked in LDS
    char* gd_lds = smem + SM::GOFF + wave * (2 * S::KD * kTileBytes) + lane * 16;
    if constexpr (FORM != kFormNoViewDirs) {
        const Ray rv = make_ray<true>(a, seed);
        h8 GDhi[S::KD], GDlo[S::KD];
        encode<2, S::KD, X3>(rv.vx, rv.vy, rv.vz, half, GDhi, GDlo);
#pragma unroll
        for (int k = 0; k < S::KD; ++k) {
            *reinterpret_cast<h8*>(gd_lds + (2 * k) * kTileBytes) = GDhi[k];
            *reinterpret_cast<h8*>(gd_lds + (2 * k + 1) * kTileBytes) = GDlo[k];
        }
    }

    FineSampler fs;
    // coarse weights, then the cdf: one buffer per wave (= per packet), or ONE for the workgroup's single packet (SPLIT), which
    // wave 0 alone writes - all four waves compute the same values - and everyone reads behind a workgroup barrier
    fs.wc = reinterpret_cast<float*>(smem + SM::WOFF) + (SPLIT ? 0 : wave * (kPacketMaxSamples * kRaysPerWave)) + (lane & 31);
    const bool wc_writer = !SPLIT || wave == 0;
    fs.stride = kRaysPerWave; fs.u_tab = s_u; fs.ns = ns; fs.ni = ni;
    fs.cd.t_tab = s_t; fs.cd.omt_tab = s_omt; fs.cd.ns = ns;
    fs.cd.jitter = a.t_rand; fs.cd.row = row;                         // training-mode forward: host-drawn random rows
    fs.u_rand = a.u_rand;
    __syncthreads();

    Composite comp;
    uint32_t flags = 0;
#ifdef NWE_STAMPS
    unsigned long long st_enc = 0, st_sync = 0, st_mlp = 0, st_comp = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
    const unsigned long long st_real = __builtin_amdgcn_s_memrealtime();   // 100 MHz: the in-kernel clock is d(memtime) / d(memrealtime) x 100 MHz
#endif
    for (int pass = 0; pass < (ni > 0 ? 2 : 1); ++pass) {
        const NetMfma& net = pass == 0 ? nc : nf;
        const float* bias = s_bias + (pass == 0 ? 0 : SM::BIAS_BYTES / 4);
        const float* dot_tab = bias + NCH * 32;
        const int Stot = pass == 0 ? ns : ns + ni;
        const float* noise = pass == 0 ? a.noise_c : a.noise_f;
        const float* raw_in = pass == 0 ? a.raw_in_c : a.raw_in_f;   // test hook: network outputs from the caller (uniform)
        if (pass == 0 && a.w_in) {                                   // test hook: coarse weights from the caller, no coarse pass
            if (wc_writer) for (int s = 0; s < ns; ++s) fs.wc[s * kRaysPerWave] = a.w_in[rclamp * ns + s];
            continue;
        }
        comp.reset();
        if constexpr (SPLIT) {
            // depths are produced strictly in order: zq[0..3] = this iteration's four samples, zq[4] = the first of the next
            int produced = 0;
            auto gen = [&](const Ray& ray) -> float {
                const int i = produced++;
                if (i >= Stot) return 0.f;
                if (pass == 0) return fs.cd.z(ray, i);
                return a.z_fine_in ? a.z_fine_in[rclamp * Stot + i] : fs.next(ray);
            };
            float zq[5], zp[4];
            {
                const Ray ray = fresh_ray();
                if (pass == 1) {
                    if (wc_writer) fs.build_cdf();     // in place: one wave, then everyone reads
                    __syncthreads();
                    fs.start(ray);
                    if (wants_survey(a.out)) {
                        const SampleSurvey sv = fs.survey(ray);
                        if (live) flags |= store_survey(a.out, ridx, sv);
                    }
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) zq[k] = gen(ray);
            }
            float4* xch = reinterpret_cast<float4*>(smem + SM::XOFF);
            const int n_it = (Stot + 3) / 4;
            // composite the (up to four) samples of iteration `it`, shaded by the four waves, in sample order
            auto drain = [&](int it) {
                const float4* x = xch + (it & 1) * (kWaves * kRaysPerWave) + (lane & 31);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int si = 4 * it + k;
                    if (si < Stot) {
                        const float w = comp.accumulate(x[k * kRaysPerWave], zp[k]);
                        if (pass == 0) {
                            if (wc_writer) fs.wc[si * kRaysPerWave] = w;
                            if (live && a.out.weights_coarse) a.out.weights_coarse[ridx * ns + si] = w;
                        }
                    }
                }
            };
            for (int it = 0; it < n_it; ++it) {
#ifdef NWE_STAMPS
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
                if (!raw_in) {
                    wk.start(net.stream, bias);
                    wk.begin(S::N_L0, 0, 0);
#pragma unroll
                    for (int i = 0; i < S::N_L0; ++i) wk.piece(i);
                    wk.begin(S::N_L0, 1, 1);
#pragma unroll
                    for (int i = 0; i < S::N_L0; ++i) wk.piece(i);
                }
                const Ray ray = fresh_ray();
                const int s_own = 4 * it + wave;
                const bool own_valid = s_own < Stot;
                float z_own = zq[0], z_nxt = zq[1];
                if (wave == 1) { z_own = zq[1]; z_nxt = zq[2]; }
                if (wave == 2) { z_own = zq[2]; z_nxt = zq[3]; }
                if (wave == 3) { z_own = zq[3]; z_nxt = zq[4]; }
                float nz[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) nz[k] = gen(ray);
                float rr, rg, rb, rs;
                if (raw_in) {
                    __syncthreads();             // publishes the previous iteration's shaded samples
                    if (it > 0) drain(it - 1);
#pragma unroll
                    for (int k = 0; k < 4; ++k) zp[k] = zq[k];
                    const float4 v = *reinterpret_cast<const float4*>(raw_in + (rclamp * Stot + (own_valid ? s_own : Stot - 1)) * 4);
                    rr = v.x; rg = v.y; rb = v.z; rs = v.w;
                } else {
                float px, py, pz;
                point_at(ray, z_own, px, py, pz);
                h8 Ghi[S::KG], Glo[S::KG];
                encode<5, S::KG, X3>(__fdiv_rn(px, 10.f), __fdiv_rn(py, 10.f), __fdiv_rn(pz, 10.f), half, Ghi, Glo);
#ifdef NWE_STAMPS
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                st_enc += t1 - t0;
#endif
                wk.template sync<false>();   // also publishes the previous iteration's shaded samples
                Frags F;
#pragma unroll
                for (int k = 0; k < PD; ++k) {
                    F.hi[k] = *reinterpret_cast<const h8*>(wk.cur() + lane * 16 + (2 * k) * kTileBytes);
                    if (X3) F.lo[k] = *reinterpret_cast<const h8*>(wk.cur() + lane * 16 + (2 * k + 1) * kTileBytes);
                }
                if (it > 0) drain(it - 1);   // behind the fragment reads, whose latency it covers
#pragma unroll
                for (int k = 0; k < 4; ++k) zp[k] = zq[k];
#ifdef NWE_STAMPS
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                wk.st_t0 = t2;
                st_sync += t2 - t1;
#endif
                mlp_eval<W, D, SKIP, X3, FORM>(wk, F, lane, net.inv_scale, Ghi, Glo, gd_lds, dot_tab, rr, rg, rb, rs);
#ifdef NWE_STAMPS
                st_mlp += __builtin_amdgcn_s_memtime() - t2;
#endif
                }
#ifdef NWE_STAMPS
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
                if (own_valid) {
                    xch[(it & 1) * (kWaves * kRaysPerWave) + wave * kRaysPerWave + (lane & 31)] =
                        Composite::shade(rr, rg, rb, rs, z_own, z_nxt, s_own + 1 == Stot, ray.dnorm, noise ? noise[rclamp * Stot + s_own] : 0.f);
                    if (lane_live) {
                        float* raw = pass == 0 ? a.out.raw_coarse : a.out.raw_fine;
                        if (raw) {
                            *reinterpret_cast<float4*>(raw + (ridx * Stot + s_own) * 4) = make_float4(rr, rg, rb, rs);
                            if (bad(rr) || bad(rg) || bad(rb) || bad(rs)) flags |= NWE_FLAG_RAW;
                        }
                        if (pass == 1 && a.out.z_fine) a.out.z_fine[ridx * Stot + s_own] = z_own;
                    }
                }
                zq[0] = zq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) zq[k + 1] = nz[k];
#ifdef NWE_STAMPS
                st_comp += __builtin_amdgcn_s_memtime() - t3;
#endif
            }
            __syncthreads();
            drain(n_it - 1);
            __syncthreads();   // the exchange buffers are free again for the next pass
        } else {
            float z_cur, z_next = 0.f;
            {
                const Ray ray = fresh_ray();
                if (pass == 0) z_cur = fs.cd.z(ray, 0);
                else {
                    fs.prepare(ray);
                    if (wants_survey(a.out)) {
                        const SampleSurvey sv = fs.survey(ray);
                        if (live) flags |= store_survey(a.out, ridx, sv);
                    }
                    z_cur = a.z_fine_in ? a.z_fine_in[rclamp * Stot] : fs.next(ray);
                }
            }
            for (int s = 0; s < Stot; ++s) {
    #ifdef NWE_STAMPS
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    #endif
                // chunks 0 and 1 (layer 0, tiles 0 and 1) fly while the sample's depth and gamma(x) are computed
                if (!raw_in) {
                    wk.start(net.stream, bias);
                    wk.begin(S::N_L0, 0, 0);
    #pragma unroll
                    for (int i = 0; i < S::N_L0; ++i) wk.piece(i);
                    wk.begin(S::N_L0, 1, 1);
    #pragma unroll
                    for (int i = 0; i < S::N_L0; ++i) wk.piece(i);
                }
                const Ray ray = fresh_ray();
                if (s + 1 < Stot) {
                    if (pass == 0) z_next = fs.cd.z(ray, s + 1);
                    else z_next = a.z_fine_in ? a.z_fine_in[rclamp * Stot + s + 1] : fs.next(ray);
                }
                float rr, rg, rb, rs;
                if (raw_in) {
                    const float4 v = *reinterpret_cast<const float4*>(raw_in + (rclamp * Stot + s) * 4);
                    rr = v.x; rg = v.y; rb = v.z; rs = v.w;
                } else {
                float px, py, pz;
                point_at(ray, z_cur, px, py, pz);
                h8 Ghi[S::KG], Glo[S::KG];
                // handler.py:93: scalar_factor = 10, a true division (embedding.py:48)
                encode<5, S::KG, X3>(__fdiv_rn(px, 10.f), __fdiv_rn(py, 10.f), __fdiv_rn(pz, 10.f), half, Ghi, Glo);
    #ifdef NWE_STAMPS
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                st_enc += t1 - t0;
    #endif
                wk.template sync<false>();
                Frags F;
    #pragma unroll
                for (int k = 0; k < PD; ++k) {
                    F.hi[k] = *reinterpret_cast<const h8*>(wk.cur() + lane * 16 + (2 * k) * kTileBytes);
                    if (X3) F.lo[k] = *reinterpret_cast<const h8*>(wk.cur() + lane * 16 + (2 * k + 1) * kTileBytes);
                }
    #ifdef NWE_STAMPS
                const unsigned long long t2 = __builtin_amdgcn_s_memtime();
                wk.st_t0 = t2;
                st_sync += t2 - t1;
    #endif
                mlp_eval<W, D, SKIP, X3, FORM>(wk, F, lane, net.inv_scale, Ghi, Glo, gd_lds, dot_tab, rr, rg, rb, rs);
    #ifdef NWE_STAMPS
                st_mlp += __builtin_amdgcn_s_memtime() - t2;
    #endif
                }
    #ifdef NWE_STAMPS
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    #endif
                const float w = comp.step(rr, rg, rb, rs, z_cur, z_next, s + 1 == Stot, ray.dnorm, noise ? noise[rclamp * Stot + s] : 0.f);
                if (pass == 0) fs.wc[s * kRaysPerWave] = w;
                if (lane_live) {
                    if (pass == 0 && a.out.weights_coarse) a.out.weights_coarse[ridx * ns + s] = w;
                    float* raw = pass == 0 ? a.out.raw_coarse : a.out.raw_fine;
                    if (raw) {
                        *reinterpret_cast<float4*>(raw + (ridx * Stot + s) * 4) = make_float4(rr, rg, rb, rs);
                        if (bad(rr) || bad(rg) || bad(rb) || bad(rs)) flags |= NWE_FLAG_RAW;
                    }
                    if (pass == 1 && a.out.z_fine) a.out.z_fine[ridx * Stot + s] = z_cur;
                }
                z_cur = z_next;
    #ifdef NWE_STAMPS
                st_comp += __builtin_amdgcn_s_memtime() - t3;
    #endif
            }
        }
        if (live) {
            flags |= store_ray(a.out, ridx, comp, pass == 1, a.white_bkgd != 0);
            if (ni == 0) flags |= store_ray(a.out, ridx, comp, true, a.white_bkgd != 0);
        }
    }
    if (flags && a.out.flags) atomicOr(a.out.flags, flags);
#ifdef NWE_STAMPS
    if (a.stamps && lane == 0) {   // diagnostic build only: a buffer no other code reads
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * kWaves + wave) * 10;
        o[0] = st_enc; o[1] = st_sync; o[2] = st_mlp; o[3] = st_comp; o[4] = __builtin_amdgcn_s_memtime() - st_begin;
        o[5] = wk.st_pre; o[6] = wk.st_wait; o[7] = wk.st_post;
        o[8] = __builtin_amdgcn_s_memrealtime() - st_real; o[9] = st_begin;
    }
#endif
}

template <int W, int D, int SKIP, int FORM>
inline void launch_one(RenderArgs a, const NetMfma& nc, const NetMfma& nf, bool three_pass, bool split, int64_t ray_first, int64_t rays,
                       hipStream_t stream) {
    if (rays <= 0) return;
    a.ray_first = ray_first;
    const int64_t per_wg = split ? kRaysPerWave : kWaves * kRaysPerWave;
    const unsigned blocks = (unsigned)((rays + per_wg - 1) / per_wg);
#define NWE_LAUNCH(X3_, SPLIT_, LEAN_) \
    hipLaunchKernelGGL((render_mfma_kernel<W, D, SKIP, X3_, SPLIT_, FORM, LEAN_>), dim3(blocks), dim3(256), 0, stream, a, nc, nf)
    const bool lean = is_lean(a);
    if (three_pass) {
        if (split) { if (lean) NWE_LAUNCH(true, true, true); else NWE_LAUNCH(true, true, false); }
        else       { if (lean) NWE_LAUNCH(true, false, true); else NWE_LAUNCH(true, false, false); }
    } else {
        if (split) { if (lean) NWE_LAUNCH(false, true, true); else NWE_LAUNCH(false, true, false); }
        else       { if (lean) NWE_LAUNCH(false, false, true); else NWE_LAUNCH(false, false, false); }
    }
#undef NWE_LAUNCH
}

template <int W, int D, int SKIP, int FORM>
bool launch_t(const RenderArgs& a, const NetMfma& nc, const NetMfma& nf, bool three_pass, int decomposition, hipStream_t stream,
                     LaunchInfo* info) {
    using S = Shape<W, D>;
    constexpr int NCH = S::n_chunks(FORM);
    if (nc.n_chunks != NCH || (a.n_importance > 0 && nf.n_chunks != NCH)) return false;   // the kernel copies NCH bias rows
    // One workgroup per CU at a time, so a launch costs (rounds of workgroups) x (sample iterations per workgroup).  Three
    // plans, same arithmetic: all packets; all sample-split (finer units, ~6 % overhead: redundant sequential part and
    // exchange); or the full rounds as packets and the ragged last round sample-split in a second launch behind it.
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    const int64_t rays_wg = kWaves * kRaysPerWave;
    const double its = (double)(a.n_samples + (a.n_importance > 0 ? a.n_samples + a.n_importance : 0));
    const double its_split = 1.06 * (double)((a.n_samples + 3) / 4 + (a.n_importance > 0 ? (a.n_samples + a.n_importance + 3) / 4 : 0));
    const auto rounds = [&](int64_t rays, int64_t per_wg) { return (double)(((rays + per_wg - 1) / per_wg + cus - 1) / cus); };
    const int64_t full = (a.n_rays / rays_wg / cus) * cus * rays_wg;            // rays in complete rounds of packet workgroups
    const double t_packet = rounds(a.n_rays, rays_wg) * its;
    const double t_split = rounds(a.n_rays, kRaysPerWave) * its_split;
    const double t_hybrid = full > 0 && full < a.n_rays ? (double)(full / rays_wg / cus) * its + rounds(a.n_rays - full, kRaysPerWave) * its_split : 1e300;
    // two launches when the model promises at least 0.8 %: the 800x800 frame (19 full rounds + 136 workgroups) is 1.0 % by the
    // model and measures -0.4 % (368.4 vs 370.0 ms, alternating, tools/plan_ab.py); a second launch costs a few microseconds
    const double t_single = t_packet <= t_split ? t_packet : t_split;
    int plan = t_hybrid < 0.992 * t_single ? 2 : (t_packet <= t_split ? 0 : 1);
    if (decomposition >= 0) plan = decomposition;   // nwe_debug_set_decomposition: tests force one
    if (a.n_samples > kPacketMaxSamples) plan = 1;  // only the single-packet workgroup has LDS for that many coarse weights
    if (info) { info->plan = plan; info->rays_first = plan == 2 ? full : a.n_rays; info->mid_recorded = false; }
    if (plan == 2) {
        launch_one<W, D, SKIP, FORM>(a, nc, nf, three_pass, false, 0, full, stream);
        if (info && info->mid) info->mid_recorded = hipEventRecord(info->mid, stream) == hipSuccess;   // the two launches timed apart
        launch_one<W, D, SKIP, FORM>(a, nc, nf, three_pass, true, full, a.n_rays - full, stream);
    } else {
        launch_one<W, D, SKIP, FORM>(a, nc, nf, three_pass, plan == 1, 0, a.n_rays, stream);
    }
    return true;
}


}  // namespace nwe
